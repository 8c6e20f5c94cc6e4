"""AWQ search traces from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_awq_trace.py

For the three model-level AWQ fixtures (tests/golden/awq_tiny_llama_{fold,self}.npz, awq_tiny_gptj_default.npz: same
models, same calibration ids, same configs) this records what the reference's two grid searches SAW, not only what they
chose: the 20-point loss history of every module tuple's scale search (awq.py:317-359) and the 10-point loss history of
every module's clip search (awq.py:436-459), captured from the reference's own `logger.debug` lines (the code is not
touched; only the logger object of its awq module is wrapped).  tests/test_gpu_models.py uses them to tell a REAL
difference (the HIP path picks a grid point the reference's own numbers rule out) from an argmin near-tie (the
reference's loss at the HIP path's grid point is within float noise of its minimum).

  awq_trace_<tag>.npz : scale_names [T] (module tuple joined by '|'), scale_hist [T,20] float64, scale_best [T] int
                        clip_names [M], clip_hist [M,10] float64, clip_best [M] int (index of the chosen ratio 1-i/100)
"""

import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402


class Tap:
    """Wraps the reference's logger: forwards nothing, remembers the messages of the two searches in order."""

    def __init__(self):
        self.scale, self.clip, self._tuple, self._module = [], [], None, None

    def info(self, msg, *a, **k):
        msg = str(msg)
        if msg.startswith("[SCALE] Processing module: "):
            self._tuple = ast.literal_eval(msg[len("[SCALE] Processing module: "):])
        elif msg.startswith("[CLIP] Processing module: "):
            self._module = msg[len("[CLIP] Processing module: "):]

    def debug(self, msg, *a, **k):
        msg = str(msg)
        if msg.startswith("The loss history of different scale:"):
            self.scale.append(("|".join(self._tuple), ast.literal_eval(msg.split(":", 1)[1])))
        elif msg.startswith("The loss history of different clip range:"):
            self.clip.append((self._module, ast.literal_eval(msg.split(":", 1)[1])))

    def warning(self, *a, **k):
        pass

    error = warning


def first_strict_min(hist):
    best, idx = float("inf"), None
    for i, v in enumerate(hist):
        if v < best:
            best, idx = v, i
    return idx


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REF)
    import transformers  # noqa: F401  (before the reference, see make_golden_models.py)
    from neural_compressor.torch.algorithms.weight_only import awq as ref_awq
    from neural_compressor.torch.quantization import AWQConfig, convert, prepare

    from tests.model_zoo import calib_ids, tiny_gptj, tiny_llama

    ids = calib_ids()
    ABSORB = {
        "fold": {"input_layernorm": ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"],
                 "post_attention_layernorm": ["mlp.gate_proj", "mlp.up_proj"],
                 "self_attn.o_proj": "self_attn.o_proj", "mlp.down_proj": "mlp.down_proj"},
        "self": {n: n for n in ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]},
    }
    cases = [(f"tiny_llama_{t}", tiny_llama, dict(absorb_layer_dict=a)) for t, a in ABSORB.items()]
    cases.append(("tiny_gptj_default", tiny_gptj, {}))
    for tag, make, kw in cases:
        tap = Tap()
        ref_awq.logger = tap
        model = make()
        model.config.use_cache = False
        cfg = AWQConfig(bits=4, group_size=32, use_sym=False, use_auto_scale=True, use_auto_clip=True, **kw)
        model = prepare(model, cfg, example_inputs=ids[0])
        for x in ids:
            model(x)
        convert(model)
        out = dict(
            scale_names=np.array([n for n, _ in tap.scale]), scale_hist=np.array([h for _, h in tap.scale], dtype=np.float64),
            scale_best=np.array([first_strict_min(h) for _, h in tap.scale], dtype=np.int64),
            clip_names=np.array([n for n, _ in tap.clip]), clip_hist=np.array([h for _, h in tap.clip], dtype=np.float64),
            clip_best=np.array([first_strict_min(h) for _, h in tap.clip], dtype=np.int64),
        )
        np.savez_compressed(os.path.join(HERE, f"awq_trace_{tag}.npz"), **out)
        print(tag, "scale searches:", len(tap.scale), "clip searches:", len(tap.clip))
        h = out["scale_hist"]
        srt = np.sort(h, axis=1)
        print("  scale: relative gap between best and runner-up:", np.round((srt[:, 1] - srt[:, 0]) / srt[:, 0], 6).tolist())
        h = out["clip_hist"]
        srt = np.sort(h, axis=1)
        print("  clip : relative gap between best and runner-up:", np.round((srt[:, 1] - srt[:, 0]) / srt[:, 0], 6).tolist())


if __name__ == "__main__":
    main()
