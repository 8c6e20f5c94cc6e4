"""Model-level golden fixtures from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_models.py

  gptq_tiny_llama.npz : prepare -> run_fn -> convert of the reference on tests/model_zoo.tiny_llama (fp32, CPU)
                        -> per-module qweight / scales / qzeros + logits of the quantised model
  rtn_tiny_llama.npz  : quantize(model, RTNConfig(...)) -> same buffers
"""

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402


def dump_modules(q, out):
    n = 0
    for name, mod in q.named_modules():
        if type(mod).__name__ == "INCWeightOnlyLinear":
            out[f"{name}.qweight"] = mod.qweight.numpy()
            out[f"{name}.qzeros"] = mod.qzeros.numpy()
            out[f"{name}.scales"] = mod.scales.numpy()
            if getattr(mod, "g_idx", None) is not None:
                out[f"{name}.g_idx"] = mod.g_idx.numpy()
            n += 1
    out["n_modules"] = np.int64(n)


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    import transformers  # noqa: F401  -- BEFORE the reference: its WOQ white list only includes transformers.Conv1D when
    #                                      transformers is already in sys.modules (torch/utils/environ.py:44-49), as in user code
    from neural_compressor.torch.quantization import GPTQConfig, RTNConfig, convert, prepare, quantize

    from tests.model_zoo import calib_ids, tiny_gpt2, tiny_gptj, tiny_llama, tiny_opt

    ids = calib_ids()

    def run_fn(model):
        for x in ids:
            model(x)

    tmp = tempfile.mkdtemp()  # GPTQConfig(model_path=<existing dir>) quirk, gptq.py:277
    for tag, kw in {
        "sym_g32": dict(bits=4, group_size=32, use_sym=True, block_size=128),
        "asym_g32": dict(bits=4, group_size=32, use_sym=False, block_size=128),
        # the option tests of the reference's test_gptq.py (act_order :137-160, use_mse_search, true_sequential :187-208)
        "act_order": dict(bits=4, group_size=32, use_sym=True, block_size=128, act_order=True),
        "true_seq": dict(bits=4, group_size=32, use_sym=True, block_size=128, true_sequential=True),
        "mse": dict(bits=4, group_size=32, use_sym=False, block_size=128, use_mse_search=True),
    }.items():
        model = tiny_llama()
        cfg = GPTQConfig(model_path=tmp, **kw)
        model = prepare(model, cfg)
        run_fn(model)
        q = convert(model)
        out = {}
        dump_modules(q, out)
        with torch.no_grad():
            out["logits"] = q(ids[0]).logits.float().numpy()
        np.savez_compressed(os.path.join(HERE, f"gptq_tiny_llama_{tag}.npz"), **out)
        print(tag, "modules:", int(out["n_modules"]))

    # AWQ with an explicit absorb_layer_dict (the reference's jit-trace discovery fails on this transformers version
    # and silently falls back to self-absorption; the explicit dict pins the structure for both implementations):
    #   "fold": norms absorb q/k/v and gate/up, o_proj / down_proj absorb themselves (MulLinear)
    #   "self": every Linear absorbs itself (what the reference's fallback produces)
    from neural_compressor.torch.quantization import AWQConfig

    ABSORB = {
        "fold": {"input_layernorm": ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"],
                 "post_attention_layernorm": ["mlp.gate_proj", "mlp.up_proj"],
                 "self_attn.o_proj": "self_attn.o_proj", "mlp.down_proj": "mlp.down_proj"},
        "self": {n: n for n in ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]},
    }
    for tag, absorb in ABSORB.items():
        model = tiny_llama()
        model.config.use_cache = False  # the reference re-runs blocks with the captured kwargs; a live KV cache would grow
        cfg = AWQConfig(bits=4, group_size=32, use_sym=False, use_auto_scale=True, use_auto_clip=True, absorb_layer_dict=absorb)
        model = prepare(model, cfg, example_inputs=ids[0])
        run_fn(model)
        q = convert(model)
        out = {}
        dump_modules(q, out)
        for name, mod in q.named_modules():
            if type(mod).__name__ == "MulLinear":
                out[f"{name}.input_scale"] = mod.input_scale.float().numpy()
            if type(mod).__name__ == "LlamaRMSNorm":
                out[f"{name}.weight"] = mod.weight.detach().float().numpy()
        with torch.no_grad():
            out["logits"] = q(ids[0]).logits.float().numpy()
            out["logits_fp"] = tiny_llama()(ids[0]).logits.float().numpy()
        np.savez_compressed(os.path.join(HERE, f"awq_tiny_llama_{tag}.npz"), **out)
        print("awq", tag, "modules:", int(out["n_modules"]))

    # GPT-J (the reference tests' own model family): GPTQ sym g32 and RTN asym g32
    model = tiny_gptj()
    model = prepare(model, GPTQConfig(model_path=tmp, bits=4, group_size=32, use_sym=True, block_size=128))
    run_fn(model)
    q = convert(model)
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
        out["logits_fp"] = tiny_gptj()(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "gptq_tiny_gptj_sym_g32.npz"), **out)
    print("gptj gptq modules:", int(out["n_modules"]))
    q = quantize(tiny_gptj(), RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "rtn_tiny_gptj_asym_g32.npz"), **out)
    print("gptj rtn modules:", int(out["n_modules"]))

    # AWQ on GPT-J with the reference's DEFAULT absorb discovery (its torch.jit trace works on this architecture):
    # ln_1 folds q/k/v/fc_in, out_proj and fc_out get a MulLinear
    model = tiny_gptj()
    model.config.use_cache = False
    model = prepare(model, AWQConfig(bits=4, group_size=32, use_sym=False, use_auto_scale=True, use_auto_clip=True), example_inputs=ids[0])
    run_fn(model)
    q = convert(model)
    out = {}
    dump_modules(q, out)
    for name, mod in q.named_modules():
        if type(mod).__name__ == "MulLinear":
            out[f"{name}.input_scale"] = mod.input_scale.float().numpy()
        if type(mod).__name__ == "LayerNorm":
            out[f"{name}.weight"] = mod.weight.detach().float().numpy()
            out[f"{name}.bias"] = mod.bias.detach().float().numpy()
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
        out["logits_fp"] = tiny_gptj()(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "awq_tiny_gptj_default.npz"), **out)
    print("awq gptj modules:", int(out["n_modules"]))

    # OPT (BASELINE config #1's architecture): RTN INT8 per-channel (config #1's algorithm) and GPTQ INT4
    q = quantize(tiny_opt(), RTNConfig(bits=8, group_size=-1, use_layer_wise=False))
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "rtn_tiny_opt_int8_pc.npz"), **out)
    print("opt rtn modules:", int(out["n_modules"]))
    model = prepare(tiny_opt(), GPTQConfig(model_path=tmp, bits=4, group_size=32, use_sym=False, block_size=128))
    run_fn(model)
    q = convert(model)
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
        out["logits_fp"] = tiny_opt()(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "gptq_tiny_opt_asym_g32.npz"), **out)
    print("opt gptq modules:", int(out["n_modules"]))

    # GPT-2: transformers.Conv1D layers (weight stored [in, out]).  RTN only: the reference's GPTQ export crashes on a
    # non-square Conv1D (`Q.t_()` leaves scale [out, G] against a [in, out] weight in quant_weight_w_scale, gptq.py:795-801:
    # "The size of tensor a (64) must match the size of tensor b (4)"), so there is no reference output to pin GPTQ to.
    q = quantize(tiny_gpt2(), RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "rtn_tiny_gpt2_asym_g32.npz"), **out)
    print("gpt2 rtn modules:", int(out["n_modules"]))

    model = tiny_llama()
    q = quantize(model, RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    out = {}
    dump_modules(q, out)
    with torch.no_grad():
        out["logits"] = q(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "rtn_tiny_llama_asym_g32.npz"), **out)
    print("rtn modules:", int(out["n_modules"]))


if __name__ == "__main__":
    main()
