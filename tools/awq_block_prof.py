"""AWQ (BASELINE config #3) on one Llama-2-7B-shaped block, twice (first run warms the libraries), with the per-phase
wall-clock of the second run.  usage: INC_MI355X_AWQ_TIMING=1 python tools/awq_block_prof.py [runs]"""
import json
import os
import sys
import time

os.environ.setdefault("INC_MI355X_AWQ_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import LlamaConfig, LlamaForCausalLM

from neural_compressor_amd.torch.quantization import AWQConfig, convert, prepare

device = "cuda"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator().manual_seed(1)
SEQ = int(os.environ.get("AWQ_PROF_SEQ", "2048"))  # 2048 = the headline calibration set (bench.py awq_block)
ids = [torch.randint(0, 32000, (1, SEQ), generator=g) for _ in range(128)]
for run in range(runs):
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device(device):
        m = LlamaForCausalLM(cfg)
    m = m.to(torch.bfloat16).eval()
    m.config.use_cache = False
    qc = AWQConfig(bits=4, group_size=128, use_sym=False, use_auto_scale=True, use_auto_clip=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        m = prepare(m, qc, example_inputs=ids[0].to(device))
        t1 = time.perf_counter()
        for x in ids:
            m(x.to(device))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        m = convert(m)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(json.dumps(dict(run=run, total_s=round(t3 - t0, 3), prepare_s=round(t1 - t0, 3), calib_forward_s=round(t2 - t1, 3),
                          convert_s=round(t3 - t2, 3), phases=getattr(m, "awq_phase_s", None))), flush=True)
    del m
    torch.cuda.empty_cache()
