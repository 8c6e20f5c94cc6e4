import ctypes, sys, time, torch
sys.path.insert(0, "/root/repo")
from neural_compressor_amd import ops
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
dev = torch.device("cuda")
src = torch.randn(65536, device=dev).to(torch.bfloat16)
sink = torch.zeros(4096 * 256, device=dev)
def rate(stream, blocks=2048, iters=400):
    with torch.cuda.stream(stream):
        ops.probe_mfma_bf16(src, sink, blocks, iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fl = 0
        for _ in range(5):
            fl += ops.probe_mfma_bf16(src, sink, blocks, iters)
        torch.cuda.synchronize()
        return fl / (time.perf_counter() - t0) / 1e12
print("default stream", round(rate(torch.cuda.current_stream()), 1), "TFLOP/s")
for name, words in [("8 words all ones", [0xffffffff] * 8), ("8 words, low 7 words", [0xffffffff] * 7 + [0]), ("8 words, each 0x0fffffff", [0x0fffffff] * 8),
                    ("1 word 0xffffffff", [0xffffffff]), ("1 word 0x0000ffff", [0x0000ffff]), ("2 words ones", [0xffffffff] * 2), ("8 words 0x55555555", [0x55555555] * 8),
                    ("8 words: first only", [0xffffffff] + [0] * 7)]:
    try:
        print(name, round(rate(masked_stream(words)), 1), "TFLOP/s")
    except Exception as e:
        print(name, "failed", e)
