"""CPU, build container only: checkpoints saved by THIS repository on the MI355X (tests/golden/ckpt/ours_*, produced by
scripts/make_our_checkpoints.py) are loaded by the UNMODIFIED reference's load() -- the drop-in claim for the on-disk formats
in the direction reference <- this repository.  Skipped where /root/reference does not exist (the GPU box)."""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CKPT = os.path.join(ROOT, "tests", "golden", "ckpt")

pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REF, "neural_compressor")) or not os.path.isdir(os.path.join(CKPT, "ours_rtn_default")),
    reason="needs the reference checkout and the committed fixtures",
)


@pytest.fixture(scope="module")
def ref_load():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import _install_stubs

    _install_stubs()
    added = REF not in sys.path
    if added:
        sys.path.insert(0, REF)
    import transformers  # noqa: F401  (before the reference, see make_golden_models.py)
    from neural_compressor.torch.quantization import load

    yield load
    if added:
        sys.path.remove(REF)


def _packed(model):
    return {n: m for n, m in model.named_modules() if type(m).__name__ == "INCWeightOnlyLinear"}


def test_reference_loads_our_default_format(ref_load):
    from tests.model_zoo import calib_ids, tiny_llama

    ours = np.load(os.path.join(CKPT, "ours_logits.npz"))
    q = ref_load(os.path.join(CKPT, "ours_rtn_default"), original_model=tiny_llama(), format="default", device="cpu")
    mods = _packed(q)
    assert len(mods) == 14
    state = torch.load(os.path.join(CKPT, "ours_rtn_default", "quantized_weight.pt"), map_location="cpu", weights_only=True)
    for n, m in mods.items():  # the reference consumed exactly the tensors this repository wrote
        assert torch.equal(m.qweight, state[n + ".qweight"]) and torch.equal(m.scales, state[n + ".scales"]) and torch.equal(m.qzeros, state[n + ".qzeros"])
    with torch.no_grad():
        y = q(calib_ids()[0]).logits.float().numpy()
    ref = ours["rtn_default"]  # MI355X, fp16 compute
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= 2e-2


def test_reference_loads_our_huggingface_format(ref_load):
    from tests.model_zoo import calib_ids

    ours = np.load(os.path.join(CKPT, "ours_logits.npz"))
    q = ref_load(os.path.join(CKPT, "ours_gptq_hf"), format="huggingface", device="cpu")
    assert len(_packed(q)) == 14
    with torch.no_grad():
        y = q(calib_ids()[0]).logits.float().numpy()
    ref = ours["gptq_hf"]
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) <= 2e-2
