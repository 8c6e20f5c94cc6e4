"""GPU: the transformers-like front-end -- quantise while loading, save_low_bit, reload, AutoAWQ
checkpoints.  Mirrors the call shapes of the reference's test/3x/torch/quantization/weight_only/test_transformers.py."""

import json
import os

import numpy as np
import pytest
import torch

import oracle.woq_oracle as O
from tests.model_zoo import calib_ids, tiny_llama

pytestmark = pytest.mark.gpu


def _woq(model):
    from neural_compressor_amd.torch.algorithms.weight_only.modules import MI355XWeightOnlyLinear

    return {n: m for n, m in model.named_modules() if isinstance(m, MI355XWeightOnlyLinear)}


def _buffers(model):
    return {n + "." + k: v.detach().cpu() for n, m in _woq(model).items() for k, v in m.state_dict().items()}


@pytest.fixture()
def float_dir(tmp_path):
    d = tmp_path / "float"
    tiny_llama(dtype=torch.float16).save_pretrained(str(d))
    return str(d)


def test_rtn_quantise_on_load_save_low_bit_reload(float_dir, tmp_path):
    from neural_compressor_amd.torch.quantization import RTNConfig, quantize
    from neural_compressor_amd.transformers import AutoModelForCausalLM, RtnConfig

    q = AutoModelForCausalLM.from_pretrained(float_dir, quantization_config=RtnConfig(bits=4, group_size=32, sym=False))
    assert len(_woq(q)) == 14 and "lm_head" not in _woq(q)
    ref = quantize(tiny_llama(dtype=torch.float16), RTNConfig(bits=4, group_size=32, use_sym=False, use_layer_wise=False))
    b, br = _buffers(q), _buffers(ref)
    assert b.keys() == br.keys()
    for k in b:
        assert torch.equal(b[k], br[k]), k
    ids = calib_ids()[0].to("cuda")
    with torch.no_grad():
        y0 = q(ids).logits.float().cpu()
    out = str(tmp_path / "low_bit")
    q.save_pretrained(out)
    files = set(os.listdir(out))
    assert {"quantize_config.json", "all_checkpoint_keys.json", "config.json"} <= files
    saved = json.load(open(os.path.join(out, "quantize_config.json")))
    assert saved["quant_method"] == "rtn" and saved["bits"] == 4 and saved["group_size"] == 32 and saved["sym"] is False
    assert "tokenizer" not in saved and "device" not in saved  # remove_redundant_parameters
    assert json.load(open(os.path.join(out, "config.json")))["quantization_config"]["quant_method"] == "rtn"
    r = AutoModelForCausalLM.from_pretrained(out)
    assert type(r.quantization_config).__name__ == "RtnConfig" and r.quantization_config.group_size == 32
    b1 = _buffers(r)
    assert b1.keys() == b.keys()
    for k in b:
        assert torch.equal(b[k], b1[k]), k
    with torch.no_grad():
        y1 = r(ids).logits.float().cpu()
    assert float((y1 - y0).norm() / y0.norm()) <= 2e-2
    # a reloaded model can be saved again (save_pretrained stays bound to save_low_bit)
    r.save_pretrained(str(tmp_path / "again"))
    assert os.path.exists(tmp_path / "again" / "quantize_config.json")


def test_gptq_front_end_equals_prepare_convert(float_dir):
    """GPTQConfig(dataset=<token tensors>) through from_pretrained == prepare / run / convert on the same rows."""
    from neural_compressor_amd.torch.quantization import GPTQConfig as TorchGPTQConfig
    from neural_compressor_amd.torch.quantization import convert, prepare
    from neural_compressor_amd.transformers import AutoModelForCausalLM, GPTQConfig

    ids = calib_ids(n=8, seq=32)
    cfg = GPTQConfig(bits=4, group_size=32, sym=True, damp_percent=0.01, desc_act=True, dataset=ids, seq_len=32,
                     n_samples=8, batch_size=1)
    # the zoo model runs eager attention; ask HF for the same kernel so both calibrations see identical activations
    q = AutoModelForCausalLM.from_pretrained(float_dir, quantization_config=cfg, attn_implementation="eager")
    import transformers

    # same float model object on both routes (HF keeps the rotary inv_freq in fp32 when it instantiates under fp16; the
    # zoo's `.to(fp16)` would round it -- last-bit activation differences that flip a handful of GPTQ codes)
    fm = transformers.AutoModelForCausalLM.from_pretrained(float_dir, dtype=torch.float16, attn_implementation="eager").eval()
    m = prepare(fm, TorchGPTQConfig(bits=4, group_size=32, use_sym=True, percdamp=0.01, act_order=True, block_size=128))
    for x in ids:
        m(x)
    m = convert(m)
    b, br = _buffers(q), _buffers(m)
    assert b.keys() == br.keys() and any(k.endswith("g_idx") for k in b)
    for k in b:
        assert torch.equal(b[k], br[k]), k
    assert not hasattr(q.quantization_config, "tokenizer") and q.quantization_config.desc_act is True
    with torch.no_grad():
        y = q(ids[0].to("cuda")).logits
    assert torch.isfinite(y).all()
    assert all(mod._plan == "fused_act_order" for mod in _woq(q).values())


def test_autoawq_checkpoint_is_repacked_on_load(tmp_path):
    """A directory in AutoAWQ's GEMM format ([K, N/8] interleaved words, config.json quantization_config
    quant_method=awq) opens with packed modules whose weights equal AutoAWQ's dequantisation (reference
    repack_awq_and_load_state_dict, transformers/quantization/utils.py:655-697)."""
    from safetensors.torch import save_file

    from neural_compressor_amd.transformers import AutoModelForCausalLM

    model = tiny_llama(dtype=torch.float16)
    gs = 32
    g = torch.Generator().manual_seed(5)
    state, want = {}, {}
    for name, p in model.state_dict().items():
        mod_name = name.rsplit(".", 1)[0]
        if name.endswith(".weight") and p.dim() == 2 and "layers." in name and "norm" not in name:
            N, K = p.shape
            aq = torch.randint(-(2**31), 2**31 - 1, (K, N // 8), generator=g, dtype=torch.int64).to(torch.int32)
            az = torch.randint(-(2**31), 2**31 - 1, (K // gs, N // 8), generator=g, dtype=torch.int64).to(torch.int32)
            sc = (torch.rand(K // gs, N, generator=g) * 0.02 + 0.004).half()
            state[mod_name + ".qweight"], state[mod_name + ".qzeros"], state[mod_name + ".scales"] = aq, az, sc
            codes = torch.from_numpy(O.awq_unpack_fields(aq.numpy())).float()
            zeros = torch.from_numpy(O.awq_unpack_fields(az.numpy())).float()
            want[mod_name] = ((codes - zeros.repeat_interleave(gs, 0)) * sc.float().repeat_interleave(gs, 0)).T  # [N, K]
        else:
            state[name] = p.contiguous()
    d = tmp_path / "awq"
    os.makedirs(d)
    cfg = model.config.to_dict()
    cfg["quantization_config"] = {"quant_method": "awq", "bits": 4, "group_size": gs, "zero_point": True, "version": "gemm"}
    json.dump(cfg, open(d / "config.json", "w"))
    save_file(state, str(d / "model.safetensors"), metadata={"format": "pt"})
    q = AutoModelForCausalLM.from_pretrained(str(d))
    mods = _woq(q)
    assert len(mods) == 14 and type(q.quantization_config).__name__ == "AwqConfig"
    for name, m in mods.items():
        got = m.recover(dtype=torch.float32).cpu()
        assert float((got - want[name]).norm() / want[name].norm()) <= 1e-3, name
    with torch.no_grad():
        y = q(calib_ids()[0].to("cuda")).logits
    assert torch.isfinite(y).all()
