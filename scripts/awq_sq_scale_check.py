"""BASELINE configs #3 (AWQ) and #4 (SmoothQuant) at full layer size on one MI355X: two Llama-2-7B-shaped blocks (AWQ) /
two Llama-2-13B-shaped blocks (SmoothQuant W8A8), random weights, synthetic calibration tokens.  Prints wall-clock per
block and sanity numbers; run through gpurun:  python scripts/awq_sq_scale_check.py [awq|sq|all]"""

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def llama(hidden, inter, heads, layers, device):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=heads, vocab_size=32000, max_position_embeddings=4096, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device(device):
        m = LlamaForCausalLM(cfg)
    m = m.to(torch.bfloat16).eval()
    m.config.use_cache = False
    return m


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    dev = torch.device("cuda", 0)
    out = {}
    g = torch.Generator().manual_seed(1)
    if what in ("awq", "all"):
        from neural_compressor_amd.torch.quantization import AWQConfig, convert, prepare

        layers, n, seq = 2, 128, 512
        model = llama(4096, 11008, 32, layers, dev)
        ids = [torch.randint(0, 32000, (1, seq), generator=g) for _ in range(n)]
        with torch.no_grad():
            ref = model(ids[0].to(dev)).logits.float()
        cfg = AWQConfig(bits=4, group_size=128, use_sym=False, use_auto_scale=True, use_auto_clip=True)
        torch.cuda.synchronize()
        t0 = time.time()
        model = prepare(model, cfg, example_inputs=ids[0].to(dev))
        for x in ids:
            model(x.to(dev))
        model = convert(model)
        torch.cuda.synchronize()
        dt = time.time() - t0
        with torch.no_grad():
            y = model(ids[0].to(dev)).logits.float()
        out["awq"] = dict(blocks=layers, samples=n, seq=seq, seconds=round(dt, 2), seconds_per_block=round(dt / layers, 2),
                          llama2_7b_estimate_s=round(32 * dt / layers, 1), rel_logit_err=round(float((y - ref).norm() / ref.norm()), 4),
                          peak_gib=round(torch.cuda.max_memory_allocated() / 2**30, 1))
        del model
        torch.cuda.empty_cache()
    if what in ("sq", "all"):
        from neural_compressor_amd.torch.algorithms.smooth_quant import W8A8Linear
        from neural_compressor_amd.torch.quantization import SmoothQuantConfig, convert, prepare

        layers, n, seq = 2, 32, 2048
        model = llama(5120, 13824, 40, layers, dev)
        ids = [torch.randint(0, 32000, (1, seq), generator=g) for _ in range(n)]
        with torch.no_grad():
            ref = model(ids[0].to(dev)).logits.float()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(ids[0].to(dev))
            torch.cuda.synchronize()
            t_fp = (time.time() - t0) / 3
        cfg = SmoothQuantConfig(alpha=0.5, folding=False, scale_sharing=True)
        cfg.set_local("lm_head", SmoothQuantConfig(w_dtype="fp32"))
        torch.cuda.synchronize()
        t0 = time.time()
        model = prepare(model, cfg, example_inputs=ids[0].to(dev))
        for x in ids:
            model(x.to(dev))
        model = convert(model)
        torch.cuda.synchronize()
        dt = time.time() - t0
        with torch.no_grad():
            y = model(ids[0].to(dev)).logits.float()
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(ids[0].to(dev))
            torch.cuda.synchronize()
            t_q = (time.time() - t0) / 3
        nq = sum(isinstance(m, W8A8Linear) for m in model.modules())
        out["smooth_quant"] = dict(blocks=layers, samples=n, seq=seq, seconds=round(dt, 2), seconds_per_block=round(dt / layers, 2),
                                   llama2_13b_estimate_s=round(40 * dt / layers, 1), w8a8_modules=nq,
                                   rel_logit_err=round(float((y - ref).norm() / ref.norm()), 4),
                                   forward_ms_bf16=round(t_fp * 1e3, 2), forward_ms_w8a8=round(t_q * 1e3, 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
