"""AWQ grid forwards: the recorded float outputs of the block's children in front of the searched Linears are replayed
instead of being recomputed (awq.py `_float_block_outputs` / `_PrefixReplay`).  Pure host logic: runs on the CPU."""
import pytest
import torch

transformers = pytest.importorskip("transformers")

from neural_compressor_amd.torch.algorithms.weight_only import awq as A


def _quantizer_with_block(n_samples=4):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=100)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).eval()
    block = model.model.layers[0]
    caught = []
    h = block.register_forward_pre_hook(lambda mod, args, kwargs: caught.append((list(args), dict(kwargs))), with_kwargs=True)
    with torch.no_grad():
        for _ in range(n_samples):
            model(torch.randint(0, 100, (1, 16)))
    h.remove()
    q = object.__new__(A.ActAwareWeightQuant)
    q.total_block_args = [a for a, _ in caught]
    q.total_block_kwargs = [k for _, k in caught]
    q._block_chunks = None
    q._float_block = None
    return q, block


@pytest.mark.parametrize("search_batch", ["1", "2"])
def test_prefix_replay_reproduces_full_forwards(monkeypatch, search_batch):
    monkeypatch.setenv("INC_MI355X_AWQ_SEARCH_BATCH", search_batch)
    q, block = _quantizer_with_block()
    with torch.no_grad():
        org = q._float_block_outputs(block)
        assert q._float_block_outputs(block) is org  # once per block
        # children behind which no leaf starts (the MLP: the block's last child) are not kept: they can never be a replayed prefix
        assert set(q._float_block["rec"]) == {"input_layernorm", "self_attn", "post_attention_layernorm"}
        cases = (
            ([block.mlp.gate_proj, block.mlp.up_proj], ["input_layernorm", "self_attn", "post_attention_layernorm"]),
            ([block.self_attn.q_proj, block.self_attn.k_proj, block.self_attn.v_proj], ["input_layernorm"]),
        )
        for mods, want in cases:
            r = q._prefix_replay(block, mods)
            assert r is not None and r.names == want
            for step in range(3):
                for m in mods:
                    m.weight.data = m.weight.data * (1.0 + 0.1 * (step + 1))
                outs = r.run(lambda: q._search_block_outputs(block))
                full = q._search_block_outputs(block)
                assert all(torch.equal(a, b) for (a, _), (b, _) in zip(outs, full))
                assert not all(torch.equal(a, b) for (a, _), (b, _) in zip(outs, org))
            r.close()
            assert not r.broken
            assert all("forward" not in getattr(block, n).__dict__ for n in want)  # the children run their own forward again


def test_prefix_replay_detects_in_place_use_of_a_recorded_output(monkeypatch):
    """A block that modifies a child's output in place cannot be replayed: the version counters give it away and the grid
    forwards fall back to the full block."""
    monkeypatch.setenv("INC_MI355X_AWQ_SEARCH_BATCH", "1")

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pre = torch.nn.Linear(8, 8)
            self.fc = torch.nn.Linear(8, 8)

        def forward(self, x):
            h = self.pre(x)
            y = self.fc(h)
            h.add_(1.0)  # in-place on the recorded output of `pre`
            return y + h

    torch.manual_seed(0)
    block = Block().eval()
    q = object.__new__(A.ActAwareWeightQuant)
    q.total_block_args = [[torch.randn(1, 4, 8)] for _ in range(3)]
    q.total_block_kwargs = [{} for _ in range(3)]
    q._block_chunks = None
    q._float_block = None
    with torch.no_grad():
        q._float_block_outputs(block)
        assert "pre" not in q._float_block["rec"]
        assert q._prefix_replay(block, [block.fc]) is None


def test_prefix_replay_can_be_switched_off(monkeypatch):
    monkeypatch.setattr(A, "PREFIX_REPLAY", False)
    q, block = _quantizer_with_block(2)
    with torch.no_grad():
        q._float_block_outputs(block)
        assert q._prefix_replay(block, [block.mlp.gate_proj, block.mlp.up_proj]) is None
