"""SmoothQuant alpha="auto" + do_blockwise golden from the UNMODIFIED reference (build container only):

    python tests/golden/make_golden_sq_auto.py   ->  tests/golden/sq_blockwise_tiny_opt.npz

TorchSmoothQuant(tiny_opt, q_func).transform(alpha="auto", folding=False, auto_alpha_args=...) on CPU (reference
smooth_quant/utility.py:1232 AutoAlpha, :2289 transform; `peft` and IPEX are stubbed -- neither is touched on this path):
  keys [G] / absorb_to_layer (json)   the absorbing modules and the Linears each feeds (the reference's jit trace works on OPT)
  layers [M], alpha_space [A], final_alpha [G]   tuned Linears, the grid, the chosen alpha per group
  final_loss [M, A]    the BLOCK loss table (every Linear of a block carries its block's losses) the final decision was taken on
  logits               of the smoothed (still floating-point) model on calib_ids()[0]
The block replay of the reference passes the block ONLY its hidden states (utility.py:1685: no attention mask) -- reproduced here.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import REF, _install_stubs  # noqa: E402

ARGS = dict(init_alpha=0.5, alpha_min=0.3, alpha_max=0.7, alpha_step=0.1, shared_criterion="max", n_samples=8, do_blockwise=True)


def main():
    sys.dont_write_bytecode = True
    _install_stubs()
    peft = types.ModuleType("peft")
    peft.PeftModel = type("PeftModel", (), {})
    sys.modules["peft"] = peft
    sys.path.insert(0, REF)
    import torch
    import transformers  # noqa: F401
    from neural_compressor.torch.algorithms.smooth_quant import utility as U

    from tests.model_zoo import calib_ids, tiny_opt

    ids = calib_ids(n=8, seq=32)
    model = tiny_opt()

    def run(m):
        for x in ids:
            m(x)

    seen = []
    orig = U.AutoAlpha._get_best_alpha

    def spy(self, absorb_to_layer, loss_alphas, shared_criterion):
        seen.append({k: {a: float(v) for a, v in d.items()} for k, d in loss_alphas.items()})
        return orig(self, absorb_to_layer, loss_alphas, shared_criterion)

    U.AutoAlpha._get_best_alpha = spy
    sq = U.TorchSmoothQuant(model, q_func=run, example_inputs=ids[0], scale_sharing=True)
    sq.transform(alpha="auto", folding=False, auto_alpha_args=dict(ARGS))
    U.AutoAlpha._get_best_alpha = orig
    import json

    keys = list(sq.absorb_to_layer.keys())
    layers = sorted(seen[-1].keys())
    space = sorted(float(a) for a in seen[-1][layers[0]].keys())
    out = dict(keys=np.array(keys), absorb_to_layer=np.array(json.dumps({k: list(v) for k, v in sq.absorb_to_layer.items()})),
               layers=np.array(layers), alpha_space=np.array(space), final_alpha=np.array([sq.alpha[k] for k in keys]),
               final_loss=np.array([[seen[-1][n][str(a)] for a in space] for n in layers], dtype=np.float64), n_decisions=np.int64(len(seen)))
    with torch.no_grad():
        out["logits"] = model(ids[0]).logits.float().numpy()
    np.savez_compressed(os.path.join(HERE, "sq_blockwise_tiny_opt.npz"), **out)
    srt = np.sort(out["final_loss"], axis=1)
    print("groups", len(keys), "layers", len(layers), "decisions", len(seen), "alphas", out["final_alpha"].tolist())
    print("relative gap best/runner-up:", np.round((srt[:, 1] - srt[:, 0]) / srt[:, 0], 5).tolist())


if __name__ == "__main__":
    main()
