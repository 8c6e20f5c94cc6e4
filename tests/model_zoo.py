"""Deterministic synthetic models shared by the golden generator and the tests (no network, no checkpoints)."""

import numpy as np
import torch


def opt125m_like(layers=12, hidden=768, ffn=3072, vocab=512, seed=0):
    """OPT-125M-shaped decoder stack (12 x {q,k,v,out_proj,fc1,fc2}, hidden 768, ffn 3072), fp32, seeded.
    Plain nn modules so that both boxes regenerate bit-identical weights from the seed."""
    g = torch.Generator().manual_seed(seed)

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(hidden, hidden)
            self.k_proj = torch.nn.Linear(hidden, hidden)
            self.v_proj = torch.nn.Linear(hidden, hidden)
            self.out_proj = torch.nn.Linear(hidden, hidden)
            self.fc1 = torch.nn.Linear(hidden, ffn)
            self.fc2 = torch.nn.Linear(ffn, hidden)

        def forward(self, x):
            a = torch.tanh(self.q_proj(x)) * torch.sigmoid(self.k_proj(x)) + self.v_proj(x)
            x = x + self.out_proj(a)
            return x + self.fc2(torch.relu(self.fc1(x)))

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = torch.nn.Embedding(vocab, hidden)
            self.layers = torch.nn.ModuleList([Block() for _ in range(layers)])
            self.lm_head = torch.nn.Linear(hidden, vocab, bias=False)

        def forward(self, ids):
            x = self.embed(ids)
            for layer in self.layers:
                x = layer(x)
            return self.lm_head(x)

    model = Model()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return model


def tiny_llama(seed=0, hidden=64, layers=2, heads=4, inter=128, vocab=128, dtype=torch.float32):
    """Random-init LlamaForCausalLM (HF architecture, tiny dims)."""
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(
        hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
        num_key_value_heads=heads, vocab_size=vocab, max_position_embeddings=256, tie_word_embeddings=False,
        attn_implementation="eager",
    )
    torch.manual_seed(seed)
    model = LlamaForCausalLM(cfg).to(dtype)
    model.eval()
    return model


def tiny_gptj(seed=0, dtype=torch.float32):
    """Random-init GPTJForCausalLM, the architecture of the reference tests' `tiny-random-GPTJForCausalLM`
    (test/torch/quantization/weight_only/test_gptq.py:32-37): parallel attention / MLP behind ONE LayerNorm, partial rotary,
    biased fc layers, lm_head with bias."""
    from transformers import GPTJConfig, GPTJForCausalLM

    cfg = GPTJConfig(n_embd=64, n_layer=2, n_head=4, rotary_dim=8, n_inner=128, vocab_size=128, n_positions=256,
                     bos_token_id=1, eos_token_id=2, attn_implementation="eager")
    torch.manual_seed(seed)
    model = GPTJForCausalLM(cfg).to(dtype)
    model.eval()
    return model


def tiny_gpt2(seed=0, dtype=torch.float32):
    """Random-init GPT2LMHeadModel: its projections are `transformers.Conv1D` (weight [in, out]), the second layer type
    the reference's weight-only algorithms accept."""
    from transformers import GPT2Config, GPT2LMHeadModel

    cfg = GPT2Config(n_embd=64, n_layer=2, n_head=4, n_inner=128, vocab_size=128, n_positions=256, bos_token_id=1,
                     eos_token_id=2, tie_word_embeddings=False, attn_implementation="eager",
                     use_cache=False)  # the reference re-runs blocks with the captured kwargs: a live KV cache would grow
    torch.manual_seed(seed)
    model = GPT2LMHeadModel(cfg).to(dtype)
    model.eval()
    return model


def tiny_opt(seed=0, dtype=torch.float32):
    """Random-init OPTForCausalLM: the architecture of BASELINE config #1 (OPT-125M): biased q/k/v/out_proj and fc1/fc2,
    LayerNorm with bias, ReLU, learned positions; `model.decoder.layers` is the transformer stack."""
    from transformers import OPTConfig, OPTForCausalLM

    cfg = OPTConfig(hidden_size=64, ffn_dim=128, num_hidden_layers=2, num_attention_heads=4, vocab_size=128,
                    max_position_embeddings=256, word_embed_proj_dim=64, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                    attn_implementation="eager", use_cache=False)
    torch.manual_seed(seed)
    model = OPTForCausalLM(cfg).to(dtype)
    model.eval()
    return model


def calib_ids(n=8, seq=32, vocab=128, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, vocab, (1, seq), generator=g) for _ in range(n)]


def digest(t):
    """Order-sensitive 64-bit digest of an integer array."""
    a = np.ascontiguousarray(t).reshape(-1).view(np.uint8).astype(np.uint64)
    idx = (np.arange(a.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return np.uint64((a * idx).sum() % np.uint64(2**61 - 1))
