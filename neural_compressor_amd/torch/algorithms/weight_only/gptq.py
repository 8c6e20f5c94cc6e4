"""GPTQ on the MI355X: Hessian accumulation on the matrix cores, the blocked column loop as HIP kernels.

Reference: neural_compressor/torch/algorithms/weight_only/gptq.py
  trace_gptq_target_blocks :68    find_layers :144
  RAWGPTQuantizer          :184   prepare_for_calibration :399, execute_quantization :568
  GPTQ                     :1089  add_batch :1111, fasterquant :1143
  Quantizer.find_params    :1501  quantize :1626
  GPTQuantizer             :1651

The orchestration (capture block-0 inputs, per block: hooks + forward over all calibration batches ->
fasterquant per Linear -> second forward with the quantised weights -> pack) is the reference's, kept in
Python.  The arithmetic is not:
  * add_batch      -> inc_gptq_hessian_accum   (bf16/f16 MFMA syrk, fp32 accumulate; one H per DISTINCT input:
                      q/k/v and gate/up see the same activations, the reference recomputes X^T X for each)
  * fasterquant    -> inc_gptq_hessian_finalize, inc_gptq_prepare_weight, inc_gptq_find_params,
                      inc_gptq_quant_block (serial 128-column chain, one lane per weight row),
                      inc_gptq_lazy_update (fp32 MFMA); the Cholesky trio is the hipSOLVER library call that
                      torch.linalg dispatches to on ROCm (SURVEY.md section 8, row K6').
  * export         -> the column loop already emits the integer codes; quant_weight_w_scale (reference
                      utility.py:483) is not needed; packing is inc_woq_pack on device.
Everything stays resident in HBM between the steps (no block.cpu() / .cpu() round trips as in :766-783).
"""

import math
import os
import time
from functools import partial

import torch
import torch.nn as nn

from .... import ops
from ....common.utils import logger
from ...utils.utility import batch_broadcastable, get_accelerator, get_model_device, set_module
from ..base_algorithm import Quantizer as INCQuantizer
from .modules import MI355XWeightOnlyLinear

try:
    import transformers

    _Conv1D = transformers.Conv1D
    SUPPORTED_LAYERS = (nn.Linear, transformers.Conv1D)
except Exception:  # pragma: no cover
    _Conv1D = None
    SUPPORTED_LAYERS = (nn.Linear,)

QBLOCK = 128  # columns per inc_gptq_quant_block launch


# ---------------------------------------------------------------------------------------------------
# model structure
# ---------------------------------------------------------------------------------------------------
def is_leaf(module):
    return next(module.children(), None) is None


def trace_gptq_target_blocks(module, module_types=(torch.nn.ModuleList, torch.nn.Sequential)):
    """Locate the transformer stack (first ModuleList / Sequential), what precedes it and what follows it."""
    blocks = {"embeddings": {}, "transformers_pre": {}, "transformers_name": "", "transformers": [], "transformers_post": {}}
    found = False
    for n, m in module.named_modules():
        if type(m) in module_types:
            if not found:
                blocks["transformers_name"] = n
                blocks["transformers"] = m
                found = True
        elif (is_leaf(m) and not found) or "Embedding" in type(m).__name__:
            blocks["embeddings"][n] = m
        elif found and n.find(blocks["transformers_name"]) == -1:
            blocks["transformers_post"]["name"] = n
            blocks["transformers_post"]["layer"] = m
    return blocks


def find_layers(module, layers=SUPPORTED_LAYERS, name=""):
    """{relative name: module} of every quantisable layer below `module`."""
    if isinstance(module, tuple(layers)):
        return {name: module}
    res = {}
    for child_name, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + child_name if name else child_name))
    return res


# ---------------------------------------------------------------------------------------------------
# Hinv = cholesky(cholesky_inverse(cholesky(H)), upper=True)  (reference gptq.py:1228-1230), restructured
# ---------------------------------------------------------------------------------------------------
CHOL_NB = 128
# the capture pass of a block (forward with the Hessian hooks; its OUTPUT is discarded, reference gptq.py:690-702) stops at the
# last hooked Linear instead of also running that Linear and whatever follows it
CAPTURE_EARLY_STOP = True


class _CaptureDone(Exception):
    """Raised by the Hessian pre-hook of the last hooked module of a capture pass: every input has been seen."""


_LOOKAHEAD_STREAMS = {}


# The solve of the block's LAST Linear(s) in forward order (Llama: down_proj, whose K = 11008 factorisation is the critical path of
# the solve phase) runs on a stream of its own and the second forward starts without it: a pre-hook on those modules makes the
# forward's stream wait for the solve when it gets there (attention + gate / up run underneath the factorisation).  Same launches,
# same operands: bit-identical results (a module attribute: tests compare it with every solve in front of the second forward).
LATE_SOLVE = True
LAYER_LOOKAHEAD = True  # mode "layer": next round's exchange under this round's solve (tests set it to compare)
SOLVE_2D = True  # modes "rows" / "sample+rows": a block's solves on disjoint rank groups (False: every solve row-sharded over all ranks)
_TRACE_RANGES = os.environ.get("INC_MI355X_TRACE_RANGES", "0") == "1"  # roctx ranges around the phases of a block (scripts/step_timeline.py)


PHASE_IDS = {"gptq.capture_forward": 2, "gptq.solve_issue": 3, "gptq.second_forward": 4, "gptq.solve_wait": 5, "gptq.pack": 6}


class _phase:
    """One phase of quantize_block for a kernel-trace timeline (scripts/step_timeline.py): a roctx range on the host side and an
    inc_trace_marker launch (Grid_Size_X = 64 * (2 * id) at the start, 64 * (2 * id + 1) at the end) on the current stream; free
    when INC_MI355X_TRACE_RANGES is unset."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _TRACE_RANGES:
            torch.cuda.nvtx.range_push(self.name)
            ops.trace_marker(2 * PHASE_IDS[self.name])

    def __exit__(self, *exc):
        if _TRACE_RANGES:
            ops.trace_marker(2 * PHASE_IDS[self.name] + 1)
            torch.cuda.nvtx.range_pop()
        return False


def _lookahead_stream(device):
    """Second stream of the column loop: one per (device, calling stream), so that solves issued on different streams (the late
    solve below) do not serialise on a shared one."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    st = _LOOKAHEAD_STREAMS.get(key)
    if st is None:
        st = _LOOKAHEAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


# Look-ahead of the factorisation (the rest of every trailing update and the top-level doubling products on a second stream, underneath
# the chain of diagonal blocks; bit-identical results, tests/test_gpu_parity.py).  OFF by default: the diagonal-block kernel needs a
# whole CU's LDS, so it queues behind the workgroups of the GEMM running beside it -- K = 11008 alone measured 16.1 ms in a good run
# and 25-33 ms in others against a steady 17.0 ms on one stream (scripts/chol_time.py; a second stream masked to 224 CUs with
# hipExtStreamCreateWithCUMask, ops.cu_masked_stream, was slower still: 30 ms), and inside a GPTQ step four factorisations and the
# column loops already share the chip.
CHOL_LOOKAHEAD = False
_CHOL_SIDE_STREAMS = {}


def _chol_side_stream(device):
    """Second stream of a factorisation: one per (device, calling stream) -- concurrent factorisations do not share one."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    st = _CHOL_SIDE_STREAMS.get(key)
    if st is None:
        st = _CHOL_SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


# large products of the factorisation as three-way bf16 splits (inc_gptq_inverse_factor flags bit 1); False: exact-fp32 MFMA products.
# A module attribute, not an environment switch: tests / scripts that compare the two forms set it.
CHOL_BF16X3 = True


@torch.no_grad()
def inverse_cholesky_upper(H, check=True):
    """Upper Cholesky factor U of H^-1 (H^-1 = U^T U) for a symmetric positive definite fp32 H [K,K] in HBM.

    The reference gets it from three LAPACK factorisations (potrf -> potri -> potrf, gptq.py:1228-1230); on the GPU
    that is rocSOLVER, whose unblocked diagonal kernels were a fifth of the whole GPTQ run (profiles/r1c).  With J the
    index reversal,   J H J = Lr Lr^T  (lower Cholesky)   =>   H = (J Lr J)(J Lr J)^T  with  J Lr J  UPPER triangular
    =>   H^-1 = (J Lr^-1 J)^T (J Lr^-1 J),  and  J Lr^-1 J  is upper triangular with a positive diagonal, i.e. it IS U.
    So ONE blocked Cholesky and ONE blocked triangular inverse replace the trio (half the flops, one rounding pass
    instead of three):
      * diagonal blocks (128x128): inc_chol_diag_block factors the block AND inverts its factor in one workgroup;
      * panel solve  L[i>j, j] = A[i>j, j] @ inv(L_jj)^T  and trailing update  A[i>j, i>j] -= L_panel L_panel^T: fp32 GEMMs,
        two-level (128 inside an outer block of CHOL_OUTER columns, one deep update per outer block, lower triangle only);
      * Lr^-1 by recursive doubling: inv([[A,0],[C,B]]) = [[A^-1,0],[-B^-1 C A^-1, B^-1]], all pairs of a level independent.
    All of it is ONE C-ABI call (inc_gptq_inverse_factor, csrc/ifac.hip + chol.hip: this library's own MFMA GEMMs); the Python +
    torch.mm form of rounds 1-3 and the rocSOLVER trio live in tests/ab_partners.py as its A/B partners.  A non-positive pivot
    raises like torch.linalg.cholesky does.  `check=False` returns (U, info) without reading `info` back (no host
    synchronisation): the caller checks it later with `raise_if_not_spd`.
    """
    assert H.dim() == 2 and H.shape[0] == H.shape[1] and H.dtype == torch.float32
    if H.is_cuda:
        # (second stream, CHOL_LOOKAHEAD: the rest of the trailing updates and the top-level doubling products run underneath the
        # chain of diagonal blocks -- same results as one stream, unstable in time: off)
        U, info = ops.gptq_inverse_factor(H.contiguous(), aux_stream=_chol_side_stream(H.device) if CHOL_LOOKAHEAD else None,
                                          flags=2 if CHOL_BF16X3 else 0)
        if not check:
            return U, info
        raise_if_not_spd(info)
        return U
    raise RuntimeError("inverse_cholesky_upper needs a HIP tensor: the factorisation is inc_gptq_inverse_factor (there is no CPU path)")


def raise_if_not_spd(info):
    """Host-side check of the status word written by inc_chol_diag_block (synchronises with the device)."""
    bad = int(info.item())
    if bad != 0:
        raise torch.linalg.LinAlgError(
            f"inverse_cholesky_upper: the matrix is not positive definite (pivot <= 0 in diagonal block {bad})")


def hybrid_order_perm(diag_H, groupsize):
    """The reference's hybrid order (Quantizer.compute_local_perms / compute_global_perm / compose_final_perm, gptq.py:1389-1461) as
    three tensor ops on the device: inside every group the columns in descending order of diag(H), the groups in descending order
    of their largest diag(H).  A column never leaves its group, so the packed module needs no g_idx."""
    K = diag_H.numel()
    G = K // groupsize
    d = diag_H[: G * groupsize].reshape(G, groupsize)
    local = torch.argsort(d, dim=1, descending=True)
    glob = torch.argsort(d.max(dim=1).values, descending=True)
    return (local[glob] + (glob * groupsize).unsqueeze(1)).reshape(-1)


# ---------------------------------------------------------------------------------------------------
# per-layer solver
# ---------------------------------------------------------------------------------------------------
class HessianAccumulator:
    """H = (2/n) sum_batches X^T X in the reference's running-mean form (gptq.py:1136-1141), fp32 [K,K] in HBM.

    One accumulator may be shared by several layers that see the same input tensor.

    Calibration batches are staged and folded in several at a time: `H <- H*n/(n+B) + (2/(n+B)) * X_B^T X_B` with X_B the
    B staged batches stacked along tokens is exactly the reference's update for a batch of size B (its `tmp = inp.shape[0]`),
    and at one 2048-token sample per launch the 256x256 syrk spends as long on the read-modify-write of H (64-462 MiB) as
    on the math (profiles/r1_pmc).  The staging copy is immediate, so later in-place edits of the activation are harmless.
    """

    # 65536: at 16384 tokens per launch a GPTQ step of the BASELINE shape was 5 % slower (318-324 vs 305 ms, A/B on one box): the
    # read-modify-write of H and the launch's last partial round are paid a quarter as often, and the model's own forward runs
    # larger GEMMs (see RAWGPTQuantizer._run_block, which stacks a stage's worth of calibration batches per forward)
    STAGE_TOKENS = int(os.environ.get("INC_MI355X_HESSIAN_STAGE_TOKENS", "65536"))
    # an input that already holds a full stage of tokens (the driver's stacked forwards) is read in place instead of being
    # copied into the staging buffer -- once the accumulator has SEEN that the model leaves such an input alone: the first
    # eligible batch is still copied, and its version counter is compared when the update is launched (after the forward);
    # only if it is unchanged do the later forwards skip the copy.  A model that edits a Linear's input in place later in
    # the same forward therefore simply keeps the copying path.  (A class attribute: tests set it to False to force the copy.)
    ZERO_COPY = True

    def __init__(self, columns, device):
        self.columns = columns
        self.device = device
        self.H = None  # allocated on the first batch: aliased layers never allocate theirs
        self._n = 0           # batches already folded into H
        self._pending = 0     # batches staged (or held by reference), not yet folded
        self._stage = None    # [capacity tokens, K] staging buffer in the activation dtype
        self._fill = 0
        self._direct = None   # (x2d, version) of an input read in place at the next launch
        self._zc_ok = None    # None: unknown yet; True: inputs survive their forward unmodified (zero-copy allowed); False: they do not
        self._probe = None    # (x2d, version) of the copied batch whose version decides _zc_ok at launch time
        self.defer = False    # True: a full stage waits for flush_many (one launch for all Hessians of a forward)
        self.finalized = None  # (Hinv, dead, perm) cache keyed by (percdamp, act_order)
        self._info = None      # status word of a factorisation whose check was deferred (see check())
        self._info_host, self._info_event = None, None  # its pinned copy and the event behind that copy (inverse_factor)
        self._handles = None   # pending broadcasts of a factor received from its owner rank (mode "sample+rows")
        self._ready = None     # event of a factorisation that ran on a side stream (prefactor)

    @property
    def nsamples(self):
        return self._n + self._pending

    def add_batch(self, inp):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x2d = inp.reshape(-1, inp.shape[-1])
        T = x2d.shape[0]
        if self.H is None:
            self.H = torch.zeros((self.columns, self.columns), dtype=torch.float32, device=x2d.device)
        if self._direct is not None:
            self.flush()
        if self.ZERO_COPY and self._pending == 0 and T >= self.STAGE_TOKENS and x2d.is_contiguous() and x2d.data_ptr() % 16 == 0:
            if self._zc_ok:
                self._direct = (x2d, x2d._version)
                self._pending = b
                if not self.defer:
                    self.flush()
                return
            if self._zc_ok is None and self.defer:
                self._probe = (x2d, x2d._version)  # copied below; its version at launch time (after the forward) decides
        if self._stage is not None and (self._stage.dtype != x2d.dtype or self._fill + T > self._stage.shape[0]):
            self.flush()
            if self._stage.dtype != x2d.dtype or T > self._stage.shape[0]:
                self._stage = None
        if self._stage is None:
            self._stage = torch.empty((max(self.STAGE_TOKENS, T), self.columns), dtype=x2d.dtype, device=x2d.device)
        self._stage[self._fill:self._fill + T].copy_(x2d)
        self._fill += T
        self._pending += b
        if self._fill >= self.STAGE_TOKENS and not self.defer:
            self.flush()

    def allreduce(self, group=None):
        """Sample-sharded calibration (neural_compressor_amd/distributed.py, mode "sample"): fold what is staged, then
        combine the per-rank running means into the global one -- H = sum_r (n_r / n) H_r -- over RCCL.  Every rank ends
        up with the same H and the same sample count, so the factorisation and the column loop that follow are identical
        on all ranks."""
        from ....distributed import allreduce_hessian

        self.flush()
        if self.H is None:  # this rank saw no batch for the layer
            self.H = torch.zeros((self.columns, self.columns), dtype=torch.float32, device=self.device)
        self.H, self._n = allreduce_hessian(self.H, self._n, group=group)

    def _launch_item(self):
        """(H, x2d, beta, alpha) of the pending update  H <- H*n/(n+B) + 2/(n+B) X_B^T X_B, or None."""
        if self._pending == 0:
            return None
        if self._direct is not None:
            x, version = self._direct
            if x._version != version:
                raise RuntimeError("a Linear's input was modified in place before its Hessian update was launched; "
                                   "set HessianAccumulator.ZERO_COPY = False for this model")
        else:
            x = self._stage[: self._fill]
            if self._probe is not None:
                px, pv = self._probe
                self._zc_ok = px._version == pv
                self._probe = None
        n = self._n + self._pending
        return self.H, x, self._n / n, 2.0 / n

    def _committed(self):
        self._n += self._pending
        self._fill, self._pending, self._direct = 0, 0, None
        if self._zc_ok and self._stage is not None and self._stage.shape[0] >= self.STAGE_TOKENS:
            self._stage = None  # full stages are read in place from now on: drop the [65536, K] copy buffer (1.4 GB at K = 11008)

    def flush(self):
        item = self._launch_item()
        if item is not None:
            ops.gptq_hessian_accum(*item)
            self._committed()

    @staticmethod
    def flush_many(accs, only_due=False):
        """Fold the pending batches of several accumulators with ONE launch (inc_gptq_hessian_accum_multi) when the
        library takes the batch, else one launch each.  Deterministic; equal to the single launches up to fp32 summation order
        (the batched launch may sum its last tiles over several token ranges, csrc/gptq.hip "tail split").  `only_due`: leave
        partially filled stages alone (batches that cannot be stacked keep accumulating up to STAGE_TOKENS)."""
        todo, seen = [], set()
        for acc in accs:
            if id(acc) in seen:
                continue
            seen.add(id(acc))
            if only_due and acc._direct is None and acc._fill < acc.STAGE_TOKENS:
                continue
            item = acc._launch_item()
            if item is not None:
                todo.append((acc, item))
        if not todo:
            return
        if not (len(todo) > 1 and ops.gptq_hessian_accum_multi([item for _, item in todo])):
            for _, item in todo:
                ops.gptq_hessian_accum(*item)
        for acc, _ in todo:
            acc._committed()

    def inverse_factor(self, percdamp, act_order, hybrid_groupsize=0):
        """Upper Cholesky factor of (H + damp I)^-1 (gptq.py:1186-1231); consumes H.  Cached so that layers
        sharing the accumulator factorise once.  `hybrid_groupsize` > 0: H is first rearranged by the reference's hybrid order
        (gptq.py:1203-1209) and the third return value is that permutation."""
        key = (float(percdamp), bool(act_order)) + ((int(hybrid_groupsize),) if hybrid_groupsize else ())
        if self.finalized is not None and self.finalized[0] == key:
            if self._handles:  # the factor is arriving from its owner rank: order this stream behind the transfers
                for h in self._handles:
                    h.wait()
                self._handles = None
            if self._ready is not None:
                # factorised on a side stream (prefactor): order THIS stream behind it.  The event stays: layers that share the
                # accumulator may be solved on different streams (the late solve), and every one of them has to wait
                torch.cuda.current_stream().wait_event(self._ready)
            return self.finalized[1:]
        self.flush()
        self._stage = None
        assert self.finalized is None, "an accumulator can only be finalised with one (percdamp, act_order) setting"
        H = self.H
        if H is None:  # no calibration data reached this layer: H = 0 -> every column is "dead" (H = I after the fix)
            H = torch.zeros((self.columns, self.columns), dtype=torch.float32, device=self.device)
        elif H.is_cuda:
            H.record_stream(torch.cuda.current_stream(H.device))  # accumulated on the main stream, consumed (and dropped) on this one
        hybrid_diag = None
        if hybrid_groupsize:
            # diag(H) as the reference sees it at gptq.py:1205 -- after the dead-column fix (H[dead, dead] = 1), before the damping
            hybrid_diag = torch.diagonal(H).clone()
            hybrid_diag[hybrid_diag == 0] = 1
        dead = ops.gptq_hessian_finalize(H, percdamp)
        perm = None
        if act_order:
            perm = torch.argsort(torch.diagonal(H), descending=True)
            H = H[perm][:, perm].contiguous()
        elif hybrid_groupsize:
            perm = hybrid_order_perm(hybrid_diag, int(hybrid_groupsize))
            H = H[perm][:, perm].contiguous()
        Hinv, self._info = inverse_cholesky_upper(H, check=False)
        # the status word goes to pinned host memory behind the factorisation, on the stream that produced it: `check` then waits for
        # THIS event only.  (`info.item()` is a copy on the caller's stream plus a synchronisation of it -- in the block loop that
        # stream already holds the whole second forward, and the chip ran dry after every block: ~1.5 ms per step.)
        self._info_host, self._info_event = None, None
        if self._info.is_cuda:
            self._info_host = torch.empty(1, dtype=self._info.dtype, pin_memory=True)
            self._info_host.copy_(self._info, non_blocking=True)
            self._info_event = torch.cuda.current_stream(self._info.device).record_event()
        self.H = None
        self.finalized = (key, Hinv, dead, perm)
        return Hinv, dead, perm

    def prefactor(self, stream, percdamp, act_order, hybrid_groupsize=0):
        """Run the factorisation on `stream` (a side stream): the independent Hessians of a block are factorised
        concurrently -- each is a chain of one-workgroup diagonal-block kernels and fp32 GEMMs that leaves most of the chip
        idle on its own -- instead of one after another.  The consumer (`inverse_factor` on the solve's stream) waits on
        the recorded event."""
        main = torch.cuda.current_stream()
        self.flush()
        H = self.H
        stream.wait_stream(main)
        with torch.cuda.stream(stream):
            if H is not None:
                H.record_stream(stream)  # allocated on `main`, read (and dropped) under `stream`
            out = self.inverse_factor(percdamp, act_order, hybrid_groupsize)
            for t in out:
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)  # allocated under `stream`, consumed by the column loop on `main`
            self._ready = stream.record_event()

    def check(self):
        """Raise if the (deferred) factorisation met a non-positive pivot; one host synchronisation."""
        if self._info is not None:
            info, self._info = self._info, None
            if self._info_event is not None:  # (a factor received from another rank has no event: read it back)
                self._info_event.synchronize()
                info, self._info_host, self._info_event = self._info_host, None, None
            try:
                raise_if_not_spd(info)
            except torch.linalg.LinAlgError as e:
                raise torch.linalg.LinAlgError(f"{e} [Hessian of {self.columns} columns, {self.nsamples} calibration batches]") from None

    # -- mode "sample+rows" (neural_compressor_amd/distributed.py) ------------------------------------------------------
    def reduce_to_owner(self, ctx, owner, n_total):
        """Sample-sharded calibration: sum the ranks' Hessians onto `owner` (H = sum_r (n_r / n) H_r, exact up to fp32
        summation order); the other ranks drop theirs."""
        self.flush()
        self._stage = None
        if self.H is None:  # this rank saw no batch for the layer
            self.H = torch.zeros((self.columns, self.columns), dtype=torch.float32, device=self.device)
        self.H.mul_(float(self._n))
        ctx.reduce(self.H, dst=owner)
        if ctx.rank == owner:
            if n_total > 0:
                self.H.div_(float(n_total))
        else:
            self.H = None
        self._n = int(n_total)

    def exchange_factor(self, ctx, owner, percdamp, act_order, hybrid_groupsize=0):
        """The owner factorises; every other rank receives (Hinv, dead, perm) by broadcast (asynchronously under RCCL:
        `inverse_factor` waits on the handles when a solve first needs the factor)."""
        if ctx.rank == owner:
            Hinv, dead, perm = self.inverse_factor(percdamp, act_order, hybrid_groupsize)
        else:
            self.flush()
            self._stage, self.H = None, None
            K = self.columns
            Hinv = torch.empty((K, K), dtype=torch.float32, device=self.device)
            dead = torch.empty(K, dtype=torch.uint8, device=self.device)
            perm = torch.empty(K, dtype=torch.int64, device=self.device) if (act_order or hybrid_groupsize) else None
            self.finalized = ((float(percdamp), bool(act_order)) + ((int(hybrid_groupsize),) if hybrid_groupsize else ()), Hinv, dead, perm)
            self._info = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the "not positive definite" status word travels with the factor: every rank raises together in check() instead of the
        # owner alone (the others would walk into the next collective and hang)
        if ctx.world == 1:  # a rank group of one (2-D form): nothing to send
            return
        handles = [ctx.broadcast(t, owner, async_op=True) for t in (Hinv, dead, perm, self._info) if t is not None]
        handles = [h for h in handles if h is not None]
        if ctx.rank != owner and handles:
            self._handles = handles


class GPTQ:
    """One Linear under GPTQ (interface of the reference's class GPTQ, gptq.py:1089)."""

    def __init__(self, layer, W=None, device="cuda", accumulator=None):
        self.layer = layer
        self.device = device
        self.is_conv1d = _Conv1D is not None and isinstance(layer, _Conv1D)
        w = layer.weight if W is None else W
        shape = w.shape
        self.rows, self.columns = (shape[1], shape[0]) if self.is_conv1d else (shape[0], shape[1])
        self.acc = accumulator or HessianAccumulator(self.columns, device)
        self.perm = None
        self.cfg = {}
        self.row_ctx = None  # distributed.CalibrationGroup: solve only this rank's weight rows, all-gather the results

    # the reference's Quantizer.configure (gptq.py:1375) copies the per-layer dict onto the quantizer
    defer_check = False  # RAWGPTQuantizer sets it and checks every factorisation of a block at once

    def configure(self, weight_config_this_layer):
        self.cfg = dict(weight_config_this_layer)

    @property
    def H(self):
        self.acc.flush()  # staged batches are part of the Hessian a caller sees
        return self.acc.H

    @property
    def nsamples(self):
        return self.acc.nsamples

    def add_batch(self, inp, out=None):
        self.acc.add_batch(inp)

    @staticmethod
    def check_hybrid_order(columns, groupsize, act_order, static_groups):
        """What hybrid_order requires (gptq.py:1203-1209, 1403-1461), checked BEFORE any factorisation or factor exchange: the block
        driver calls it when the solvers are configured, so every rank raises at the same point and no Hessian is factorised for a
        solve that cannot run."""
        assert not act_order, "Error: hybrid_act_order is not allowed with act_order"  # (gptq.py:1204)
        if static_groups:
            raise NotImplementedError("hybrid_order with static_groups: the reference looks the groups' parameters up by PERMUTED "
                                      "position there (gptq.py:1273-1277 take idx from act_order's perm only)")
        if groupsize == -1 or (groupsize < columns and columns % int(groupsize) != 0):
            raise ValueError("hybrid_order needs a group size that divides the number of columns (the reference's permutation covers "
                             "columns // groupsize whole groups, gptq.py:1403-1461)")

    lookahead = True  # the column loop's second stream (bit-identical either way; tests compare)

    def column_loop(self, w32, Hinv, scale, zero, loop_scale, loop_zero, codes, Q, gs, kernel_gs, blocksize, bits, sym, dynamic_groups, mse):
        """The loop of gptq.py:1250-1304 as ONE C-ABI call: per 128 columns `[find_params] -> chain -> next-128 update` on this
        stream and the rest of the trailing update on a second stream underneath the next chain (look-ahead: every column still
        receives its updates in block order from the same 128-column tiles, so W, the codes and Q are bit-identical to the
        one-stream loop).  tests/ab_partners.python_column_loop issues the same launches one by one from Python (its A/B partner)."""
        N, K = w32.shape
        flags = (ops.GPTQ_DYNAMIC_GROUPS if dynamic_groups else 0) | (ops.GPTQ_MSE if mse else 0)
        look = self.lookahead and K % QBLOCK == 0 and blocksize % QBLOCK == 0 and K >= 3 * QBLOCK and w32.is_cuda
        side = _lookahead_stream(w32.device) if look else None
        err_ws = torch.empty((2, N, QBLOCK), dtype=torch.float32, device=w32.device)
        if side is not None:
            for t in (w32, Hinv, scale, zero, loop_scale, loop_zero, codes, Q, err_ws):
                t.record_stream(side)
        ops.gptq_quantize_layer(w32, Hinv, scale, zero, None if loop_scale is scale else loop_scale,
                                None if loop_scale is scale else loop_zero, codes, Q, err_ws, gs, kernel_gs, blocksize, bits, sym, flags,
                                aux_stream=side)

    def fasterquant(self, W, blocksize=128, percdamp=0.01, groupsize=-1, act_order=False, hybrid_order=False,
                    fp8_aware=False, static_groups=False):
        """Returns (scale [N,G] fp32, scale_bf16_to_fp8, zero [N,G] fp32, Q [weight shape, weight dtype]);
        the integer codes (uint8 [N,K], original column order) are left in `self.codes`."""
        if fp8_aware:
            raise NotImplementedError("fp8_aware (INT4 weights pre-scaled for Gaudi's fp8 matrix units) is a Gaudi W4A8 option outside the MI355X scope")
        if hybrid_order:
            self.check_hybrid_order(self.columns, groupsize, act_order, static_groups)
        bits = int(self.cfg.get("bits", 4))
        sym = bool(self.cfg.get("sym", False))
        mse = bool(self.cfg.get("mse", False))  # GPTQConfig(use_mse_search=True): shrink-grid search in find_params
        if self.cfg.get("dtype", "int") != "int" or self.cfg.get("use_double_quant", False):
            raise NotImplementedError("GPTQ on MI355X quantises to plain integer formats")
        weight_shape, weight_dtype = W.shape, W.dtype
        if self.is_conv1d:
            W = W.t()
        W = W.contiguous()
        N_all, K = W.shape
        tick = time.time()
        ctx = self.row_ctx
        if ctx is not None:
            # every op below is row-wise given Hinv (gptq.py:1250-1304): this rank solves rows [r0, r1) of the (stacked)
            # weight; codes / Q / scale / zero of all ranks are all-gathered at the end (collective C2)
            from ....distributed import row_shard

            r0, r1, shard = row_shard(N_all, ctx.rank, ctx.world)
            W = W[r0:r1].contiguous()
        N = W.shape[0]

        gs = K if (groupsize == -1 or groupsize >= K) else int(groupsize)
        Hinv, dead, perm = self.acc.inverse_factor(percdamp, act_order, gs if hybrid_order else 0)
        G = math.ceil(K / gs)
        scale = torch.empty((N, G), dtype=torch.float32, device=W.device)
        zero = torch.empty((N, G), dtype=torch.float32, device=W.device)

        if N == 0:  # this rank owns no row of a small layer: it only takes part in the all-gather below
            w32 = torch.empty((0, K), dtype=torch.float32, device=W.device)
        elif groupsize == -1:
            # per-channel parameters come from W before dead columns are zeroed (gptq.py:1180-1189 order)
            w32 = ops.gptq_prepare_weight(W, None)
            ops.gptq_find_params(w32, 0, K, 1, bits, sym, scale, zero, 0, mse=mse)
            del w32
        if N > 0:
            w32 = ops.gptq_prepare_weight(W, dead)
        if static_groups and N > 0:
            ops.gptq_find_params(w32, 0, gs, G, bits, sym, scale, zero, 0, mse=mse)
        loop_scale, loop_zero = scale, zero
        if act_order:
            w32 = w32[:, perm].contiguous()
            self.perm = perm.clone()
            if static_groups and groupsize != -1:
                # the column at permuted position p keeps the parameters of its ORIGINAL group perm[p] // groupsize
                # (gptq.py:1273-1277): hand the column loop one (scale, zero) per column, i.e. group size 1
                col_group = torch.div(perm, gs, rounding_mode="floor")
                loop_scale = scale[:, col_group].contiguous()
                loop_zero = zero[:, col_group].contiguous()
        elif hybrid_order and N > 0:
            w32 = w32[:, perm].contiguous()  # whole groups, rearranged inside: the loop's dynamic groups are the reference's

        codes = torch.empty((N, K), dtype=torch.uint8, device=W.device)
        Q = torch.empty((N, K), dtype=weight_dtype, device=W.device)
        dynamic_groups = groupsize != -1 and not static_groups
        kernel_gs = gs if groupsize != -1 else 0
        if loop_scale is not scale:
            kernel_gs = 1
        blocksize = int(blocksize) if blocksize and blocksize > 0 else K
        if N > 0:
            self.column_loop(w32, Hinv, scale, zero, loop_scale, loop_zero, codes, Q, gs, kernel_gs, blocksize, bits, sym, dynamic_groups, mse)
        logger.debug("fasterquant %dx%d issued in %.3fs", N, K, time.time() - tick)

        if ctx is not None:
            codes = ctx.all_gather_rows(codes, N_all, shard)
            Q = ctx.all_gather_rows(Q, N_all, shard)
            scale = ctx.all_gather_rows(scale, N_all, shard)
            zero = ctx.all_gather_rows(zero, N_all, shard)
        if act_order:
            invperm = torch.argsort(perm)
            Q = Q[:, invperm].contiguous()
            codes = codes[:, invperm].contiguous()
        elif hybrid_order:
            # gptq.py:1320-1328: the columns back in place, the groups' parameters back in the groups' original order
            invperm = torch.argsort(perm)
            Q = Q[:, invperm].contiguous()
            codes = codes[:, invperm].contiguous()
            inv_global = torch.argsort(torch.div(perm.reshape(G, gs)[:, 0], gs, rounding_mode="floor"))
            scale = scale[:, inv_global].contiguous()
            zero = zero[:, inv_global].contiguous()
        self.codes = codes
        # with static groups the parameters belong to contiguous ORIGINAL-order groups: the packed module needs no g_idx.
        # (The reference returns only the last group's scale in this mode (:1341-1345) and its export then indexes past
        # it (utility.py:522); the [N, G] table is what that code means to produce.)
        self.export_perm = None if (static_groups or not act_order) else self.perm
        if self.is_conv1d:
            Q = Q.t().contiguous()
        Q = Q.reshape(weight_shape)
        if not self.defer_check:
            # a non-positive-definite Hessian raises here like the reference's torch.linalg.cholesky (gptq.py:1228), after all of the
            # solve's work is queued: one host synchronisation per solve.  The block driver defers it to one check per block.
            self.acc.check()
        return scale, torch.tensor([-1]), zero, Q

    def free(self):
        self.acc = None
        self.codes = None


# ---------------------------------------------------------------------------------------------------
# whole-model driver
# ---------------------------------------------------------------------------------------------------
def _to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, device) for o in obj)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    return obj


class RAWGPTQuantizer(object):
    """Block-by-block GPTQ driver (reference gptq.py:184)."""

    def __init__(self, model, weight_config={}, nsamples=128, use_max_length=True, max_seq_length=2048, device=None,
                 use_layer_wise=False, model_path="", quant_lm_head=False, dataloader=None, use_block_wise=False,
                 *args, **kwargs):
        self.model = model
        self.gptq_related_blocks = trace_gptq_target_blocks(self.model)
        self.dtype = next(iter(self.model.parameters())).dtype
        self.weight_config = weight_config
        self.quant_lm_head = quant_lm_head
        self.check_layer_config()
        self.device = torch.device(get_accelerator(device or "auto").current_device_name())
        self.is_ready = False
        # layer-wise / block-wise modes stream weights from disk to save host RAM: irrelevant with 288 GB HBM
        self.use_layer_wise = False
        self.use_block_wise = False
        self.use_max_length = use_max_length
        self.max_seq_length = max_seq_length
        self.nsamples = nsamples
        self.share_hessians = kwargs.get("share_hessians", True)
        self.factor_streams = int(os.environ.get("INC_MI355X_GPTQ_FACTOR_STREAMS", "4"))
        self._fstreams = []
        self._late_stream = None  # stream of the block's last solve (LATE_SOLVE)
        self._block_writes_input = None  # decided by the first block forward (see _run_block)
        self._stacks = {}  # first batch index of a forward group -> (stacked hidden states, the list entries that are its slices)
        self.block_callback = kwargs.get("block_callback", None)  # used by the multi-GPU driver
        # sample-sharded multi-GPU calibration: every rank feeds ITS calibration samples through prepare()/run_fn and the
        # per-layer Hessians are all-reduced before each solve; pass True (default process group) or a process group
        self.hessian_allreduce = kwargs.get("hessian_allreduce", None)
        # mode "sample+rows" / "rows" (distributed.py): the i-th distinct Hessian of a block is factorised on rank
        # i % world only and its factor broadcast, every solve is row-sharded and all-gathered.  Pass True (default
        # process group) or a process group; together with hessian_allreduce the Hessians are REDUCED to the owner
        # (samples sharded), without it every rank is expected to have seen all samples.
        self.row_shard_solve = kwargs.get("row_shard_solve", None)
        # 2-D form of the row-sharded solve (distributed.plan_solves_2d): the independent solves of a block run side by side on disjoint
        # rank groups (module x rows) instead of one after the other on all ranks (rows alone); same bits either way
        self.solve_2d = kwargs.get("solve_2d", SOLVE_2D)
        mode = os.environ.get("INC_MI355X_GPTQ_MULTI_GPU", "")
        if mode and (self.hessian_allreduce is None and self.row_shard_solve is None):
            import torch.distributed as dist

            from ....distributed import live as _dist_live

            live = _dist_live()
            assert mode in ("sample", "rows", "sample+rows", "layer"), f"INC_MI355X_GPTQ_MULTI_GPU={mode!r}"
            self.hessian_allreduce = live and mode in ("sample", "sample+rows")
            self.row_shard_solve = live and mode in ("rows", "sample+rows")
            if mode == "layer" and kwargs.get("independent_blocks") is None:
                kwargs["independent_blocks"] = True
        # mode "layer" (distributed.py; BASELINE north_star / SURVEY 8(e) mode B): one transformer block per rank, calibrated on the
        # FLOAT model's activations.  True = default process group (or a single process), or pass a process group.
        self.independent_blocks = kwargs.get("independent_blocks", None)
        self.layer_ctx = None
        if self.independent_blocks:
            import torch.distributed as dist

            assert not (self.hessian_allreduce or self.row_shard_solve), "independent_blocks excludes the sample / rows modes"
            if dist.is_available() and dist.is_initialized():
                from ....distributed import SINGLE_RANK_GROUP, CalibrationGroup

                ctx = CalibrationGroup(None if self.independent_blocks is True else self.independent_blocks)
                self.layer_ctx = ctx if (ctx.world > 1 or SINGLE_RANK_GROUP) else None
        self.dist_ctx = None
        if self.row_shard_solve:
            from ....distributed import SINGLE_RANK_GROUP, CalibrationGroup

            self.dist_ctx = CalibrationGroup(None if self.row_shard_solve is True else self.row_shard_solve)
            if self.dist_ctx.world == 1 and not SINGLE_RANK_GROUP:
                self.dist_ctx = None

    # -- config handling (reference :330-398) --------------------------------------------------------
    _DEFAULTS = dict(
        dtype="int", bits=4, group_size=128, block_size=128, percdamp=0.01, sym=False, act_order=False,
        hybrid_order=False, fp8_aware=False, static_groups=False, true_sequential=False, perchannel=True, mse=False,
        use_double_quant=False, double_quant_dtype="int", double_quant_bits=4, double_quant_group_size=128,
        double_quant_sym=False,
    )

    def check_layer_config(self):
        for layer_name, config in self.weight_config.items():
            for key, default in self._DEFAULTS.items():
                config[key] = config.get(key, default)
            if config["dtype"] != "int" and "int" in config["dtype"]:
                config["bits"] = int(config["dtype"].lstrip("int"))
                config["dtype"] = "int"

    def get_layer_config(self, layer_name):
        import re

        config = self.weight_config.get(layer_name, None)
        if config is not None:
            return config
        for pattern, cfg in self.weight_config.items():
            if re.compile(pattern).findall(layer_name):
                return cfg
        return None

    def get_full_layer_name(self, sub_layer_name, block_idx):
        return ".".join([self.gptq_related_blocks["transformers_name"], str(block_idx), sub_layer_name])

    @staticmethod
    def track_hidden_states(data):
        if isinstance(data, torch.Tensor):
            return data
        if isinstance(data, (tuple, list)):
            return data[0]
        return data

    # -- calibration capture (reference :399-482) ------------------------------------------------------
    @torch.no_grad()
    def prepare_for_calibration(self):
        self.cache_key_arguments = {"batch_num": 0}
        self.cache_positional_arguments = []
        self.is_ready = True
        quantizer = self

        def forward(layer, *args, **kwargs):
            quantizer.cache_key_arguments["batch_num"] += 1
            for arg, val in kwargs.items():
                if isinstance(val, torch.Tensor) or arg in ["alibi", "position_embeddings"]:
                    quantizer.cache_key_arguments.setdefault(arg, []).append(val)
            for idx, item in enumerate(args):
                if idx + 1 > len(quantizer.cache_positional_arguments):
                    quantizer.cache_positional_arguments.append([])
                quantizer.cache_positional_arguments[idx].append(item)
            raise ValueError  # the reference's control flow: abort the model forward after block 0's inputs are seen

        # the whole model lives in HBM on MI355X (no per-block host<->device shuttling)
        self.model.to(self.device)
        first_block = self.gptq_related_blocks["transformers"][0]
        self.forward_cache = first_block.forward
        first_block.forward = partial(forward, first_block)
        self.orig_model_forward_cache = self.model.forward
        model_forward_cache = self.model.forward
        device = self.device

        def model_forward(model, *args, **kwargs):
            try:
                model_forward_cache(*_to_device(args, device), **_to_device(kwargs, device))
            except ValueError:
                pass

        self.model.forward = partial(model_forward, self.model)

    @torch.no_grad()
    def remove_prepare_for_calibration(self):
        self.model.forward = self.orig_model_forward_cache
        self.gptq_related_blocks["transformers"][0].forward = self.forward_cache
        logger.info("GPTQ quantization prepared.")

    def gather_single_batch_from_dict(self, data_dict, idx):
        return {k: v[idx] for k, v in data_dict.items()}

    def gather_single_batch_from_list(self, data_list, idx):
        return [item[idx] for item in data_list]

    def find_true_sequential_config(self):
        for cfg in self.weight_config.values():
            if cfg.get("true_sequential", None) is not None:
                return cfg["true_sequential"]
        return False

    @staticmethod
    def analyze_true_sequential(module):
        layers = list(find_layers(module))
        if "q" in layers[0].lower() and "k" in layers[0].lower():
            qkv, post = [layers[0]], layers[1:]
        else:
            qkv, post = layers[0:3], layers[3:]
        return [qkv] + [[layer] for layer in post]

    # -- block forward over every calibration batch ----------------------------------------------------
    def _run_block(self, block, on_output=None, capture=False):
        """One forward of `block` over every cached calibration batch (reference :690-702 / :749-762).

        Cached batches that differ only in their hidden states (same shape, leading dimension 1, every other argument
        the same tensor values) are stacked `forward_batch` at a time (INC_MI355X_GPTQ_FORWARD_BATCH, default "auto" = one Hessian stage
        of tokens, 32 batches of 2048 tokens; 1 = one batch per forward as the reference does): a decoder block treats the rows of a stacked input independently, the
        running-mean Hessian update is the same sum either way (`add_batch` counts the leading dimension, gptq.py:1117),
        and the per-batch outputs are handed on as slices.  What changes is the size of the GEMMs the model's own
        forward runs (M = 65536 instead of 2048 at the BASELINE calibration shape) and 32x fewer elementwise launches."""
        batch_num = self.cache_key_arguments.pop("batch_num")
        in_kwargs = "hidden_states" in self.cache_key_arguments
        for group in self._forward_groups(batch_num, in_kwargs, block):
            j0 = group[0]
            kw = self.gather_single_batch_from_dict(self.cache_key_arguments, j0)
            pos = self.gather_single_batch_from_list(self.cache_positional_arguments, j0)
            src = self.cache_key_arguments["hidden_states"] if in_kwargs else self.cache_positional_arguments[0]
            if len(group) > 1:
                # the cached inputs of a group are normally the slices of ONE tensor -- the previous block's stacked output
                # (recorded below): reuse it instead of concatenating 8 x 16 MiB again for every forward
                held = self._stacks.get(j0)
                if held is not None and len(held[1]) == len(group) and all(src[j] is v for j, v in zip(group, held[1])):
                    stacked = held[0]
                else:
                    stacked = torch.cat([src[j] for j in group], dim=0)
                    self._stacks[j0] = (stacked, [src[j] for j in group])  # valid while the list still holds these objects
                if in_kwargs:
                    kw["hidden_states"] = stacked
                else:
                    pos[0] = stacked
            # The hidden states handed to the block ARE the cached calibration inputs (or the stacked tensor whose slices they
            # are): a block that writes its input in place would corrupt them for the next forward.  The first forward of the run
            # therefore gets a clone whose version counter tells; such a model then always gets clones.
            hkey = "hidden_states" if in_kwargs else None
            given = kw[hkey] if in_kwargs else pos[0]
            if isinstance(given, torch.Tensor) and self._block_writes_input is not False:
                probe_in = given.clone()
                v0 = probe_in._version
                if in_kwargs:
                    kw[hkey] = probe_in
                else:
                    pos[0] = probe_in
            else:
                probe_in = None
            try:
                out = self.track_hidden_states(block(*pos, **kw))
            except _CaptureDone:
                if not capture:
                    raise
                out = None  # a capture pass that ended at its last hooked module: there is no output, and none is wanted
            if probe_in is not None and self._block_writes_input is None:
                self._block_writes_input = probe_in._version != v0
                if self._block_writes_input:
                    logger.warning("GPTQ: the block modifies its input in place; calibration inputs are cloned for every forward")
            if on_output is not None:
                if out is None:
                    on_output(j0, None)
                elif len(group) == 1:
                    on_output(j0, out)
                else:
                    views = [out[i : i + 1] for i in range(len(group))]
                    for j, v in zip(group, views):
                        on_output(j, v)
                    if out.is_contiguous() and all(src[j] is v for j, v in zip(group, views)):
                        self._stacks[j0] = (out, views)  # the callback stored the slices: they ARE the next stacked input
        self.cache_key_arguments["batch_num"] = batch_num

    def _forward_groups(self, batch_num, in_kwargs, block=None):
        """[[batch indices sharing one forward]]; computed once (only the hidden states change from block to block).

        Batches are stacked only when that is provably the same computation: same shapes, every other argument the same
        values AND broadcastable over the batch (leading dimension 1 -- batch-folded arguments like `alibi`
        [batch*heads, 1, T] are not), and the first stacked forward of `block` reproduces the per-batch outputs (guards
        against blocks that are not row-independent, e.g. MoE routing with capacity limits).  Anything else runs one
        batch per forward, exactly like the reference."""
        cached = getattr(self, "_fgroups", None)
        if cached is not None and cached[0] == batch_num:
            return cached[1]
        fb_env = os.environ.get("INC_MI355X_GPTQ_FORWARD_BATCH", "auto")
        if fb_env == "auto":  # as many batches as make one Hessian stage (65536 tokens: 32 samples of 2048 tokens), at most 64
            h0 = (self.cache_key_arguments["hidden_states"] if in_kwargs else self.cache_positional_arguments[0])[0] if batch_num > 0 else None
            tokens = int(h0.numel() // h0.shape[-1]) if isinstance(h0, torch.Tensor) and h0.dim() >= 2 else 0
            fb = max(1, min(64, HessianAccumulator.STAGE_TOKENS // tokens)) if tokens > 0 else 8
        else:
            fb = max(1, int(fb_env))

        def same(a, b):
            if a is b:
                return True
            if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
                return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b))
            if isinstance(a, (tuple, list)) and isinstance(b, (tuple, list)) and len(a) == len(b):
                return all(same(x, y) for x, y in zip(a, b))
            return not isinstance(a, torch.Tensor) and not isinstance(b, torch.Tensor) and a == b

        def hidden(j):
            return self.cache_key_arguments["hidden_states"][j] if in_kwargs else self.cache_positional_arguments[0][j]

        def compatible(i, j):
            hi, hj = hidden(i), hidden(j)
            if not (isinstance(hi, torch.Tensor) and isinstance(hj, torch.Tensor)) or hi.shape != hj.shape or hi.shape[0] != 1 or hi.dim() < 2:
                return False
            for k, v in self.cache_key_arguments.items():
                if k != "hidden_states" and not (same(v[i], v[j]) and batch_broadcastable(v[i])):
                    return False
            return all(same(lst[i], lst[j]) and batch_broadcastable(lst[i])
                       for lst in self.cache_positional_arguments[(0 if in_kwargs else 1):])

        groups, cur = [], [0] if batch_num > 0 else []
        for j in range(1, batch_num):
            if fb > 1 and len(cur) < fb and compatible(cur[0], j):
                cur.append(j)
            else:
                groups.append(cur)
                cur = [j]
        if cur:
            groups.append(cur)
        multi = next((g for g in groups if len(g) > 1), None)
        if multi is not None and block is not None:
            # whether a block of this class treats the rows of a stacked input independently is decided once per run and group size
            # (mode "layer" regroups every round: the check is two forwards of the group)
            verdicts = self.__dict__.setdefault("_faithful_verdicts", {})
            vkey = (type(block), len(multi))
            if vkey not in verdicts:
                verdicts[vkey] = self._stacking_is_faithful(block, multi, in_kwargs)
            faithful = verdicts[vkey]
        else:
            faithful = True
        if not faithful:
            logger.warning("GPTQ: a stacked forward of this block does not reproduce its per-batch outputs; "
                           "running one calibration batch per forward")
            groups = [[j] for j in range(batch_num)]
        self._fgroups = (batch_num, groups)
        return groups

    def _stacking_is_faithful(self, block, group, in_kwargs):
        """One stacked forward of `group` against the per-batch forwards (no hooks are installed at this point of the
        first block; outputs are compared up to GEMM-shape rounding)."""
        hooks = [m._forward_hooks for m in block.modules()] + [m._forward_pre_hooks for m in block.modules()]
        saved = [dict(h) for h in hooks]
        for h in hooks:
            h.clear()  # the Hessian hooks of the caller must not see these probe forwards
        try:
            j0 = group[0]
            kw = self.gather_single_batch_from_dict(self.cache_key_arguments, j0)
            pos = self.gather_single_batch_from_list(self.cache_positional_arguments, j0)
            src = self.cache_key_arguments["hidden_states"] if in_kwargs else self.cache_positional_arguments[0]
            singles = []
            for j in group:
                if in_kwargs:
                    kw["hidden_states"] = src[j]
                else:
                    pos[0] = src[j]
                singles.append(self.track_hidden_states(block(*pos, **kw)).float())
            stacked_in = torch.cat([src[j] for j in group], dim=0)
            if in_kwargs:
                kw["hidden_states"] = stacked_in
            else:
                pos[0] = stacked_in
            try:
                out = self.track_hidden_states(block(*pos, **kw)).float()
            except Exception as e:  # shape errors of batch-folded arguments the structural test did not recognise
                logger.warning("GPTQ: stacked block forward failed (%s)", e)
                return False
            ref = torch.cat(singles, dim=0)
            if out.shape != ref.shape:
                return False
            tol = 1e-4 if stacked_in.dtype == torch.float32 else 1e-2  # 16-bit GEMM-shape noise of a block is ~3-4e-3
            return bool((out - ref).norm() <= tol * ref.norm().clamp_min(1e-30))
        finally:
            for h, sv in zip(hooks, saved):
                h.update(sv)

    def _exchange_factors(self, distinct):
        """Mode "sample+rows": combine the sharded Hessians on their owner ranks, factorise there, broadcast the factors.
        Every rank walks the accumulators in the same order (collectives match up); a rank factorises ITS Hessian right
        before broadcasting it and has already posted the receives of the cheaper ones ahead of it."""
        ctx = self.dist_ctx
        if self.hessian_allreduce:
            for acc, _, _, _ in distinct:
                acc.flush()
            counts = torch.tensor([float(acc._n) for acc, _, _, _ in distinct], dtype=torch.float64,
                                  device="cpu" if ctx.backend == "gloo" else self.device)
            counts = ctx.all_reduce(counts).tolist()  # a handful of scalars
            for i, (acc, _, _, _) in enumerate(distinct):
                acc.reduce_to_owner(ctx, ctx.owner(i), int(round(counts[i])))
        for i, (acc, percdamp, act_order, hyb) in enumerate(distinct):
            acc.exchange_factor(ctx, ctx.owner(i), percdamp, act_order, hyb)

    # -- modes "rows" / "sample+rows", 2-D form (distributed.plan_solves_2d) -------------------------------------------------------
    def _plan_2d(self, batches, solvers, distinct):
        """[(names, acc, CalibrationGroup of the batch's ranks or None when this rank is not in it, leader rank), ...] for the block's
        solves, or None when the 2-D form does not apply (switched off; a world of one; a Hessian that serves several batches -- then
        its factor would have to reach several groups, and the rows-only form keeps that simple)."""
        from ....distributed import plan_solves_2d, subgroup

        ctx = self.dist_ctx
        if not self.solve_2d or ctx.world < 2 or len(batches) < 2:
            return None
        accs = [solvers[names[0]].acc for names in batches]
        if len({id(a) for a in accs}) != len(accs) or len(accs) != len(distinct):
            return None
        shapes = [(sum(solvers[n].rows for n in names), solvers[names[0]].columns) for names in batches]
        ranks = plan_solves_2d(shapes, ctx.world)
        return [(names, acc, subgroup(r, ctx.group), r[0]) for names, acc, r in zip(batches, accs, ranks)]

    def _exchange_factors_2d(self, plan, distinct):
        """Every Hessian is reduced to its batch's leader, factorised there and broadcast INSIDE the batch's rank group only; the other
        ranks drop theirs.  Same collectives in the same order on every rank."""
        ctx = self.dist_ctx
        settings = {id(acc): (percdamp, act_order, hyb) for acc, percdamp, act_order, hyb in distinct}
        if self.hessian_allreduce:
            for _, acc, _, _ in plan:
                acc.flush()
            counts = torch.tensor([float(acc._n) for _, acc, _, _ in plan], dtype=torch.float64,
                                  device="cpu" if ctx.backend == "gloo" else self.device)
            counts = ctx.all_reduce(counts).tolist()
            for i, (_, acc, _, leader) in enumerate(plan):
                acc.reduce_to_owner(ctx, leader, int(round(counts[i])))
        for _, acc, sub, _ in plan:
            if sub is None:  # not this rank's solve: nothing of it is needed here until the results arrive
                acc.flush()
                acc._stage, acc.H = None, None
                continue
            percdamp, act_order, hyb = settings[id(acc)]
            acc.exchange_factor(sub, 0, percdamp, act_order, hyb)  # local rank 0 of the group = the leader

    def _solve_2d(self, plan, batches, solvers, layers, run_solve, scatter):
        """Phase 1: every rank runs the solve(s) of ITS group, rows sharded inside the group (all-gathered there).  Phase 2: each
        batch's leader publishes the batch -- codes, Q, scales, zero points, the act_order permutation and the factorisation's status
        word -- to the world, in batch order; every rank ends with the same quantised block, bit-identical to the rows-only form."""
        ctx = self.dist_ctx
        stacked = {}
        for i, (names, acc, sub, _) in enumerate(plan):
            if sub is None:
                continue
            for n in names:
                solvers[n].row_ctx = sub if sub.world > 1 else None
            stacked[i] = run_solve(names)
        for i, (names, acc, sub, leader) in enumerate(plan):
            sv = solvers[names[0]]
            cfg = sv.cfg
            if sub is not None:
                scale, zp, Q, codes, perm = stacked[i]
                info = acc._info if acc._info is not None else torch.zeros(1, dtype=torch.int32, device=self.device)
            else:
                rows, K = sum(solvers[n].rows for n in names), sv.columns
                gs_ = cfg["group_size"]
                G = 1 if (gs_ == -1 or gs_ >= K) else math.ceil(K / int(gs_))
                w0 = layers[names[0]].weight.data
                Q = torch.empty(w0.shape if len(names) == 1 else (rows, K), dtype=w0.dtype, device=self.device)
                scale = torch.empty((rows, G), dtype=torch.float32, device=self.device)
                zp = torch.empty((rows, G), dtype=torch.float32, device=self.device)
                codes = torch.empty((rows, K), dtype=torch.uint8, device=self.device)
                perm = (torch.empty(K, dtype=torch.int64, device=self.device)
                        if (cfg["act_order"] and not cfg["static_groups"]) else None)
                info = torch.zeros(1, dtype=torch.int32, device=self.device)
            Q = Q.contiguous()
            for t in (codes, Q, scale, zp, perm, info):
                if t is not None:
                    ctx.broadcast(t, leader)
            if sub is None or ctx.rank != leader:
                # the status word of a factorisation this rank did not run (or received inside the group): every rank raises together
                acc._info, acc._info_host, acc._info_event = info, None, None
            scatter(names, (scale, zp, Q, codes, perm))

    # -- the main loop (reference :568-887) ----------------------------------------------------------------
    @torch.no_grad()
    def execute_quantization(self, means=None, stds=None):
        true_sequential = self.find_true_sequential_config()
        blocks = self.gptq_related_blocks["transformers"]
        seq_map = self.analyze_true_sequential(blocks[0])
        for p in self.model.parameters():
            p.requires_grad = False
        if self.independent_blocks:
            self._execute_independent_blocks(blocks, seq_map if true_sequential else None)
            if self.quant_lm_head:
                self.quantize_post_layer()
            logger.info("Quantization done")
            return self.model
        for block_idx in range(len(blocks)):
            t0 = time.time()
            block = blocks[block_idx].to(self.device)
            self.quantize_block(block, block_idx, seq_map if true_sequential else None)
            if self.block_callback is not None:
                self.block_callback(block_idx, block)
            logger.info("Quantized block %d / %d in %.2fs", block_idx + 1, len(blocks), time.time() - t0)
        if self.quant_lm_head:
            self.quantize_post_layer()
        logger.info("Quantization done")
        return self.model

    # -- mode "layer": one transformer block per GPU, calibrated on the float model's activations ---------------------
    def _hidden_list(self):
        return self.cache_key_arguments["hidden_states"] if "hidden_states" in self.cache_key_arguments else self.cache_positional_arguments[0]

    def _set_hidden_list(self, lst):
        if "hidden_states" in self.cache_key_arguments:
            self.cache_key_arguments["hidden_states"] = lst
        else:
            self.cache_positional_arguments[0] = lst

    @torch.no_grad()
    def _execute_independent_blocks(self, blocks, seq_map):
        """BASELINE's north_star / SURVEY 8(e) mode B: block b is owned by rank b % world and calibrated on the inputs the FLOAT
        model gives it -- NOT on the outputs of the quantised block b-1 as the reference does (gptq.py:749-762; the documented
        deviation of this opt-in mode).  Every block's solve is then independent of every other block's, so `world` blocks are
        quantised at once.  Per block the result is what `quantize_block` gives on the same inputs (bit-identical to a single
        process run of this mode: test_gpu_models.py).  One round = `world` consecutive blocks:
          1. every rank forwards ITS calibration samples through the round's float blocks, keeping each block's inputs;
          2. the inputs of block b travel to its owner: by default point-to-point (each rank sends its shard of block b's inputs to
             rank b % world only -- over xGMI's per-pair links the round is a balanced all-to-all), or, with
             INC_MI355X_GPTQ_ACT_EXCHANGE=broadcast, every shard is broadcast to all ranks and kept by the owner;
          3. each rank quantises its block on the full calibration set (Hessian forward + solve + pack; no second forward: nothing
             downstream reads the quantised block's outputs in this mode).
        At the end the packed blocks are broadcast from their owners, so every rank holds the whole quantised model."""
        self.independent_setup()
        for start in range(0, len(blocks), self._layer_state["world"]):
            self.independent_round(blocks, start, seq_map)
        self.independent_finish(blocks)

    def independent_setup(self):
        """Sample counts of every rank and the shape / dtype of one calibration sample (once per run)."""
        ctx = self.layer_ctx
        world, rank = (ctx.world, ctx.rank) if ctx is not None else (1, 0)
        exchange = os.environ.get("INC_MI355X_GPTQ_ACT_EXCHANGE", "scatter")
        assert exchange in ("scatter", "broadcast"), f"INC_MI355X_GPTQ_ACT_EXCHANGE={exchange!r}"
        n_local = self.cache_key_arguments["batch_num"]
        counts = [n_local]
        if ctx is not None:
            t = torch.zeros(world, dtype=torch.int64, device="cpu" if ctx.backend == "gloo" else self.device)
            t[rank] = n_local
            counts = [int(v) for v in ctx.all_reduce(t).tolist()]
        ref = self._hidden_list()[0] if n_local else None
        if ctx is not None:  # shape / dtype of one calibration sample, from the first rank that has any
            import torch.distributed as dist

            meta = [(tuple(ref.shape), str(ref.dtype).replace("torch.", "")) if ref is not None else None]
            src = next(r for r in range(world) if counts[r] > 0)
            dist.broadcast_object_list(meta, src=ctx._global(src), group=ctx.group)
            shape, dtype = meta[0][0], getattr(torch, meta[0][1])
        else:
            shape, dtype = tuple(ref.shape), ref.dtype
        # The exchange ships one message per (block, source rank) sized `count x shape[1:]` and slices it back into samples: that is
        # only the cached data when EVERY cached hidden state is one sample of exactly this shape (run_fn fed batches of one, all of
        # one sequence length).  Checked on every rank BEFORE any collective of a round is posted, and all ranks raise together.
        hidden = self._hidden_list()[:n_local]
        bad = int(any((not isinstance(h, torch.Tensor)) or tuple(h.shape) != tuple(shape) or h.shape[0] != 1 or h.dtype != dtype for h in hidden))
        if ctx is not None:
            flag = torch.tensor([bad], dtype=torch.int64, device="cpu" if ctx.backend == "gloo" else self.device)
            bad = int(ctx.all_reduce(flag).item())
        if bad:
            raise ValueError("layer-per-GPU calibration (independent_blocks) needs calibration batches of ONE sample each, all of the same "
                             f"shape {tuple(shape)} / dtype {dtype}: {bad} rank(s) cached something else (batch size > 1 or ragged "
                             "sequence lengths); feed run_fn one padded sample per forward, or use the sample-sharded mode")
        self._layer_state = dict(world=world, rank=rank, exchange=exchange, counts=counts, n_total=sum(counts), shape=shape, dtype=dtype)

    @torch.no_grad()
    def independent_round(self, blocks, start, seq_map=None):
        """One round of mode "layer": blocks start .. start + world - 1, one per rank (steps 1-3 of `_execute_independent_blocks`)."""
        from ....distributed import owner_of_block

        st = self._layer_state
        ctx, world, rank = self.layer_ctx, st["world"], st["rank"]
        t0 = time.time()
        timing = st.get("timing")  # {"forward_s", "exchange_s", "quantize_s"} accumulated when the caller put a dict there (bench.py)

        def mark(key, since):
            if timing is None:
                return since
            torch.cuda.synchronize()
            now = time.perf_counter()
            timing[key] = timing.get(key, 0.0) + now - since
            return now

        tp = mark("_", time.perf_counter()) if timing is not None else 0.0

        def rounds_blocks(first):
            return list(range(first, min(first + world, len(blocks))))

        def forwards(rb):
            # float forwards of this rank's samples; block b's inputs are kept (the list entries are replaced, not overwritten)
            kept = {}
            for b in rb:
                blocks[b].to(self.device)
                kept[b] = list(self._hidden_list())
                if b + 1 < len(blocks) or self.quant_lm_head:
                    def replace(j, out):
                        self._hidden_list()[j] = out

                    self._run_block(blocks[b], on_output=replace)
            return kept

        def post(rb, kept):
            mine_ = next((b for b in rb if owner_of_block(b, world) == rank), None)
            if ctx is None:
                return dict(start=rb[0], mine=mine_, full=kept[mine_])
            pend = self._post_block_inputs(ctx, rb, kept, st["counts"], st["shape"], st["dtype"], st["exchange"], mine_)
            pend.update(start=rb[0], mine=mine_)
            return pend

        round_blocks = rounds_blocks(start)
        # 1. + 2. this round's inputs: posted by the previous call (look-ahead) or produced now
        pend = st.pop("prefetched", None)
        if pend is not None and pend["start"] != start:
            # the look-ahead forwards of ANOTHER round have already advanced this rank's hidden states and its messages are posted:
            # recomputing here would calibrate on the wrong activations.  Rounds must be walked in order.
            st["prefetched"] = pend
            raise RuntimeError(f"independent_round({start}): the exchange of round {pend['start']} is already posted (look-ahead); "
                               "rounds must be called in order, or independent_finish() first")
        if pend is None:
            pend = post(round_blocks, forwards(round_blocks))
        tp = mark("forward_s", tp)
        # look-ahead (LAYER_LOOKAHEAD): the NEXT round's float forwards run now and its block inputs are
        # posted (batch_isend_irecv, asynchronous under RCCL) before this round's block is quantised, so they cross xGMI underneath the
        # quantisation instead of in front of it.  Same forwards on the same data, same messages: identical results.
        if ctx is not None and LAYER_LOOKAHEAD and start + world < len(blocks) and st["exchange"] == "scatter":
            nxt = rounds_blocks(start + world)
            st["prefetched"] = post(nxt, forwards(nxt))
            tp = mark("lookahead_forward_s", tp)  # (the NEXT round's forwards: its own key, not part of this round's exchange time)
        mine = pend["mine"]
        full = pend["full"] if "full" in pend else self._finish_block_inputs(ctx, pend, st["counts"])
        del pend
        tp = mark("exchange_s", tp)
        # 3. quantise the own block on the whole calibration set
        if mine is not None:
            saved = (self._hidden_list(), self.cache_key_arguments["batch_num"], getattr(self, "_fgroups", None), self._stacks,
                     {k: v for k, v in self.cache_key_arguments.items() if k not in ("hidden_states", "batch_num")},
                     [lst for lst in self.cache_positional_arguments])
            try:
                self._view_all_samples(full, st["n_total"])
                self.quantize_block(blocks[mine], mine, seq_map, propagate=False)
                if self.block_callback is not None:
                    self.block_callback(mine, blocks[mine])
            finally:
                hidden, batch_num, fgroups, stacks, kws, poss = saved
                for k, v in kws.items():
                    self.cache_key_arguments[k] = v
                for i, lst in enumerate(poss):
                    self.cache_positional_arguments[i] = lst
                self._set_hidden_list(hidden)
                self.cache_key_arguments["batch_num"] = batch_num
                self._fgroups, self._stacks = fgroups, stacks
        del full
        mark("quantize_s", tp)
        logger.info("Quantized blocks %d..%d of %d (one per rank) in %.2fs", round_blocks[0] + 1, round_blocks[-1] + 1, len(blocks), time.time() - t0)

    def independent_finish(self, blocks):
        """Every rank ends with the whole quantised model: the owners broadcast their packed blocks."""
        from ....distributed import owner_of_block

        ctx = self.layer_ctx
        self._drain_prefetched()
        if ctx is not None:
            for b in range(len(blocks)):
                self._broadcast_packed_block(ctx, blocks[b], owner_of_block(b, ctx.world))

    def _drain_prefetched(self):
        """Wait for (and drop) a look-ahead exchange that no round consumed -- a run that stops after some rounds (bench.py times K
        rounds) must not leave posted point-to-point operations and their buffers behind."""
        st = getattr(self, "_layer_state", None)
        pend = st.pop("prefetched", None) if st else None
        if pend is not None and "full" not in pend and self.layer_ctx is not None:
            self._finish_block_inputs(self.layer_ctx, pend, st["counts"])

    def _view_all_samples(self, hidden, n_total):
        """Point the calibration cache at `hidden` (n_total samples): every other cached argument is the first local batch's,
        repeated -- this mode requires what stacking requires anyway: arguments that do not differ between batches."""
        n_local = self.cache_key_arguments["batch_num"]
        for k, v in list(self.cache_key_arguments.items()):
            if k in ("hidden_states", "batch_num"):
                continue
            assert n_local > 0, "a rank without calibration samples cannot own a block (it has no attention mask / position ids to reuse)"
            self.cache_key_arguments[k] = [v[0]] * n_total
        first = 0 if "hidden_states" in self.cache_key_arguments else 1
        for i in range(first, len(self.cache_positional_arguments)):
            self.cache_positional_arguments[i] = [self.cache_positional_arguments[i][0]] * n_total
        self._set_hidden_list(list(hidden))
        self.cache_key_arguments["batch_num"] = n_total
        self._fgroups, self._stacks = None, {}

    def _exchange_block_inputs(self, ctx, round_blocks, kept, counts, shape, dtype, exchange, mine):
        """Collective C3 of SURVEY 8: returns the inputs of this rank's block as a list of n_total [1, seq, hidden] tensors in global
        sample order (rank 0's samples first), or None when this rank owns no block of the round.  Shards travel as ONE message per
        (block, source rank): 2 GiB / world for Llama-2-7B at 128 x 2048 tokens."""
        pend = self._post_block_inputs(ctx, round_blocks, kept, counts, shape, dtype, exchange, mine)
        return pend["full"] if "full" in pend else self._finish_block_inputs(ctx, pend, counts)

    def _post_block_inputs(self, ctx, round_blocks, kept, counts, shape, dtype, exchange, mine):
        """First half of the exchange: post every send / receive of the round.  Returns {"full": ...} when the exchange completed here
        (broadcast form), else the pending state `_finish_block_inputs` waits on."""
        import torch.distributed as dist

        from ....distributed import owner_of_block

        world, rank = ctx.world, ctx.rank
        staged = ctx.backend == "gloo"
        dev = "cpu" if staged else self.device

        def shard_of(b):
            if not kept[b]:
                return torch.empty((0,) + tuple(shape[1:]), dtype=dtype, device=dev)
            x = torch.cat(kept[b], dim=0)
            return x.cpu() if staged else x

        parts = None
        if mine is not None:
            parts = [None] * world
            parts[rank] = kept[mine]
        if exchange == "broadcast":
            scratch = {}
            for b in round_blocks:
                owner = owner_of_block(b, world)
                for s in range(world):
                    if counts[s] == 0:
                        continue
                    if s == rank:
                        buf = shard_of(b)
                    else:
                        key = counts[s]
                        if owner == rank or key not in scratch:
                            buf = torch.empty((counts[s],) + tuple(shape[1:]), dtype=dtype, device=dev)
                            if owner != rank:
                                scratch[key] = buf  # non-owners receive into a reused buffer and drop it
                        else:
                            buf = scratch[key]
                    dist.broadcast(buf, src=ctx._global(s), group=ctx.group)
                    if owner == rank and s != rank:
                        buf = buf.to(self.device) if staged else buf
                        parts[s] = [buf[i : i + 1] for i in range(counts[s])]
            return dict(full=None if parts is None else [x for s in range(world) if counts[s] > 0 for x in parts[s]])
        ops_, recvs, keep = [], {}, []
        for b in round_blocks:
            owner = owner_of_block(b, world)
            if owner == rank:
                for s in range(world):
                    if s != rank and counts[s] > 0:
                        buf = torch.empty((counts[s],) + tuple(shape[1:]), dtype=dtype, device=dev)
                        recvs[s] = buf
                        ops_.append(dist.P2POp(dist.irecv, buf, ctx._global(s), ctx.group))
            elif counts[rank] > 0:
                x = shard_of(b)
                keep.append(x)
                ops_.append(dist.P2POp(dist.isend, x, ctx._global(owner), ctx.group))
        works = dist.batch_isend_irecv(ops_) if ops_ else []
        return dict(works=works, recvs=recvs, keep=keep, parts=parts, staged=staged)

    def _finish_block_inputs(self, ctx, pend, counts):
        """Second half: wait for the posted messages and assemble the owner's sample list."""
        for w in pend["works"]:
            w.wait()
        if not pend["staged"]:
            torch.cuda.current_stream().synchronize()
        parts = pend["parts"]
        for s, buf in pend["recvs"].items():
            buf = buf.to(self.device) if pend["staged"] else buf
            parts[s] = [buf[i : i + 1] for i in range(counts[s])]
        pend["keep"].clear()
        if parts is None:
            return None
        return [x for s in range(ctx.world) if counts[s] > 0 for x in parts[s]]

    def _broadcast_packed_block(self, ctx, block, owner):
        """The packed modules of `block` from rank `owner` to every rank (C2's role in this mode: ~110 MiB per Llama-2-7B block)."""
        import torch.distributed as dist

        meta = [None]
        if ctx.rank == owner:
            meta[0] = []
            for name, m in block.named_modules():
                if isinstance(m, MI355XWeightOnlyLinear):
                    ctor = dict(in_features=m.in_features, out_features=m.out_features, dtype=m.dtype, bits=m.bits, group_size=m.group_size,
                                zp=hasattr(m, "qzeros"), bias=getattr(m, "bias", None) is not None, g_idx=getattr(m, "g_idx", None) is not None,
                                use_optimum_format=m.use_optimum_format, compression_dim=m.compression_dim,
                                compression_dtype=m.compression_dtype, scale_dtype=m.scale_dtype)
                    bufs = [(bn, tuple(t.shape), t.dtype) for bn, t in sorted(m.named_buffers(recurse=False))]
                    meta[0].append((name, ctor, bufs))
        dist.broadcast_object_list(meta, src=ctx._global(owner), group=ctx.group)
        for name, ctor, bufs in meta[0]:
            if ctx.rank == owner:
                mod = dict(block.named_modules())[name]
            else:
                mod = MI355XWeightOnlyLinear(device=self.device, **ctor)
            for bn, shape, dt in bufs:
                if ctx.rank == owner:
                    t = getattr(mod, bn).contiguous()
                else:
                    t = torch.empty(shape, dtype=dt, device=self.device)
                ctx.broadcast(t, owner)
                if ctx.rank != owner:
                    setattr(mod, bn, t)  # a registered buffer name keeps the tensor as a buffer
            if ctx.rank != owner:
                mod._plan_key = None
                set_module(block, name, mod)

    @torch.no_grad()
    def quantize_post_layer(self):
        """Step 2.7 (reference :887-1080): GPTQ on the layer that follows the transformer stack (lm_head).

        Like the reference, the layer is calibrated on the LAST BLOCK'S OUTPUTS as cached (:937-941) -- the final norm
        that sits between them in the real model is not applied; restated as is so the packed lm_head is the same."""
        post = self.gptq_related_blocks["transformers_post"]
        if not post:
            logger.warning("quant_lm_head=True but no layer follows the transformer stack")
            return
        full, layer = post["name"], post["layer"]
        cfg = self.get_layer_config(full)
        if cfg is None or not isinstance(layer, SUPPORTED_LAYERS):
            logger.warning("%s can be quantized but excluded from quantization configs.", full)
            return
        logger.info("Quantizing post transformer layers")
        layer.to(self.device)
        solver = GPTQ(layer, device=self.device)
        solver.defer_check = True  # checked below, after packing is queued
        solver.configure(cfg)
        handle = layer.register_forward_hook(lambda _, inp, out: solver.add_batch(inp[0].detach()))
        for j in range(self.cache_key_arguments["batch_num"]):
            if "hidden_states" in self.cache_key_arguments:
                layer(self.cache_key_arguments["hidden_states"][j])
            else:
                layer(self.cache_positional_arguments[0][j])
        handle.remove()
        if self.dist_ctx is not None:
            self._exchange_factors([(solver.acc, cfg["percdamp"], cfg["act_order"])])
            solver.row_ctx = self.dist_ctx
        elif self.hessian_allreduce:
            solver.acc.allreduce(None if self.hessian_allreduce is True else self.hessian_allreduce)
        elif self.layer_ctx is not None and self.layer_ctx.world > 1:
            # mode "layer": every rank holds only ITS samples' last-block outputs here.  The ranks' running means are combined
            # (H = sum_r (n_r / n) H_r over RCCL) so that every rank factorises the same Hessian and packs the same lm_head, as
            # independent_finish promises; a rank without samples contributes H = 0 with weight 0.
            solver.acc.allreduce(self.layer_ctx.group)
        scale, _, zp, Q = solver.fasterquant(
            layer.weight.data, blocksize=cfg["block_size"], percdamp=cfg["percdamp"], groupsize=cfg["group_size"],
            act_order=cfg["act_order"], hybrid_order=cfg["hybrid_order"], fp8_aware=cfg["fp8_aware"],
            static_groups=cfg["static_groups"],
        )
        if isinstance(layer, nn.Linear):
            in_features, out_features = layer.in_features, layer.out_features
        else:
            in_features, out_features = layer.weight.shape[0], layer.weight.shape[1]
        zero = None if cfg["sym"] else zp
        new_module = MI355XWeightOnlyLinear(
            in_features, out_features, dtype=cfg["dtype"], bits=cfg["bits"], group_size=cfg["group_size"],
            zp=zero is not None, bias=layer.bias is not None, g_idx=solver.export_perm is not None, device=self.device,
        )
        new_module.pack_codes(solver.codes, scale, zero, layer.bias, g_idx=solver.export_perm)
        solver.acc.check()
        solver.free()
        set_module(self.model, full, new_module)

    @torch.no_grad()
    def quantize_block(self, block, block_idx, seq_map=None, propagate=True):
        sub_layers = find_layers(block)
        sequentials = seq_map if seq_map else [list(sub_layers.keys())]
        for sequential in sequentials:
            layers = {}
            for name in sequential:
                full = self.get_full_layer_name(name, block_idx)
                if self.get_layer_config(full) is None:
                    logger.warning("%s can be quantized but excluded from quantization configs.", full)
                else:
                    layers[name] = sub_layers[name]
            # Step 2.2: one GPTQ object per layer (reference :650-668)
            solvers = {}
            for name, layer in layers.items():
                solvers[name] = GPTQ(layer, device=self.device)
                solvers[name].defer_check = True  # checked once per group of solves below
                solvers[name].configure(self.get_layer_config(self.get_full_layer_name(name, block_idx)))
                cfg_ = solvers[name].cfg
                if cfg_.get("hybrid_order", False):  # before the capture pass, the factorisations and (multi-GPU) the factor exchange
                    GPTQ.check_hybrid_order(solvers[name].columns, cfg_.get("group_size", -1), cfg_.get("act_order", False), cfg_.get("static_groups", False))
            # Step 2.3: hooks feeding the Hessians (reference :670-688).  Layers that receive the very same input
            # tensor in a forward (q/k/v, gate/up) share one accumulator instead of recomputing X^T X.
            live, alias = {}, {}
            share = self.share_hessians

            fired, seen, capture = [], set(), {"probe": True, "stop": None}

            def make_hook(name):
                def body(inp):
                    x = inp[0].detach()  # (not `.data`: the version counter must stay shared, see HessianAccumulator.ZERO_COPY)
                    key = (x.data_ptr(), tuple(x.shape), x.dtype, x._version)
                    if share and key in live:
                        owner = live[key][0]
                        if alias.setdefault(name, owner) != owner:
                            raise RuntimeError(f"{name}: input sharing changed between calibration batches")
                        return
                    if alias.get(name) is not None:
                        raise RuntimeError(f"{name}: input sharing changed between calibration batches")
                    live[key] = (name, x)  # keeps x alive so its address cannot be recycled inside this forward
                    solvers[name].add_batch(x)

                def hook(_, inp):  # a forward PRE-hook: the input is all the Hessian needs (the reference hooks the output side)
                    if capture["probe"]:
                        fired.append(name)
                    seen.add(name)
                    body(inp)
                    # only once EVERY hooked module of this forward has been served (a block whose modules run in a
                    # data-dependent order simply never stops early)
                    if capture["stop"] == name and len(seen) == len(layers):
                        raise _CaptureDone

                return hook

            handles = [layers[n].register_forward_pre_hook(make_hook(n)) for n in layers]
            accs = [sv.acc for sv in solvers.values()]
            for acc in accs:
                acc.defer = True  # the Hessian updates of one forward go out as ONE launch, after that forward

            def after_forward(j, out):
                live.clear()
                seen.clear()
                HessianAccumulator.flush_many(accs, only_due=True)
                if capture["probe"]:
                    # the first (complete) forward of the pass showed the order in which the hooked modules run: when each ran
                    # exactly once, later forwards of this pass end at the last one -- its own GEMM and everything behind it
                    # only produce the output that a capture pass throws away
                    capture["probe"] = False
                    if CAPTURE_EARLY_STOP and len(fired) == len(layers) and set(fired) == set(layers):
                        capture["stop"] = fired[-1]

            with _phase("gptq.capture_forward"):
                self._run_block(block, on_output=after_forward, capture=True)
                HessianAccumulator.flush_many(accs)
            for acc in accs:
                acc.defer = False
            for h in handles:
                h.remove()
            def hybrid_gs(sv):  # group size of a hybrid-order solve (it shapes the factor: gptq.py:1203-1209), else 0
                if not sv.cfg.get("hybrid_order", False):
                    return 0
                gs_ = sv.cfg.get("group_size", -1)
                return sv.columns if (gs_ == -1 or gs_ >= sv.columns) else int(gs_)

            for name, owner in alias.items():
                if (
                    solvers[name].columns == solvers[owner].columns
                    and solvers[name].cfg["percdamp"] == solvers[owner].cfg["percdamp"]
                    and solvers[name].cfg["act_order"] == solvers[owner].cfg["act_order"]
                    and hybrid_gs(solvers[name]) == hybrid_gs(solvers[owner])
                ):
                    solvers[name].acc = solvers[owner].acc
                else:  # pragma: no cover - different damping per layer: cannot share the factorisation
                    raise RuntimeError(f"{name} shares its input with {owner} but not its GPTQ settings; pass share_hessians=False")
            distinct, seen = [], set()
            for name in layers:  # same order on every rank
                acc = solvers[name].acc
                if id(acc) not in seen:
                    seen.add(id(acc))
                    distinct.append((acc, solvers[name].cfg["percdamp"], solvers[name].cfg["act_order"], hybrid_gs(solvers[name])))
            # Linears that share one Hessian (q/k/v, gate/up) and one GPTQ setting are stacked along N and solved in ONE pass (Step 2.4)
            _KEYS = ("bits", "sym", "group_size", "block_size", "percdamp", "act_order", "hybrid_order", "fp8_aware",
                     "static_groups", "mse", "dtype", "use_double_quant")
            batches, index = [], {}
            for name, layer in layers.items():
                sv = solvers[name]
                key = None
                if isinstance(layer, nn.Linear) and self.share_hessians:
                    key = (id(sv.acc), layer.weight.dtype, tuple(sv.cfg.get(k) for k in _KEYS))
                if key is not None and key in index:
                    batches[index[key]].append(name)
                else:
                    if key is not None:
                        index[key] = len(batches)
                    batches.append([name])
            plan2d = None
            if self.dist_ctx is not None:
                plan2d = self._plan_2d(batches, solvers, distinct)
                if plan2d is not None:
                    self._exchange_factors_2d(plan2d, distinct)
                else:
                    self._exchange_factors(distinct)
                    for sv in solvers.values():
                        sv.row_ctx = self.dist_ctx
            elif self.hessian_allreduce:
                group = None if self.hessian_allreduce is True else self.hessian_allreduce
                for acc, _, _, _ in distinct:  # one all-reduce per DISTINCT accumulator
                    acc.allreduce(group)
            if self.dist_ctx is None and len(distinct) > 1 and self.factor_streams > 1:
                # the independent factorisations of the block run concurrently, the largest first (it is the critical path)
                if len(self._fstreams) < self.factor_streams:
                    # (default priority: high-priority HIP streams made the whole step 20 % SLOWER, 346 -> 424 ms)
                    self._fstreams = [torch.cuda.Stream(device=self.device) for _ in range(self.factor_streams)]
                order = sorted(range(len(distinct)), key=lambda i: -distinct[i][0].columns)
                for slot, i in enumerate(order):
                    acc, percdamp, act_order, hyb = distinct[i]
                    acc.prefactor(self._fstreams[slot % len(self._fstreams)], percdamp, act_order, hyb)
            # Step 2.4: solve (reference :690-747).  The column loop treats every weight ROW independently, so Linears
            # that share one Hessian (q/k/v, gate/up) and one GPTQ setting are stacked along N and solved in ONE pass:
            # a third of the serial 128-column steps and three times the rows in flight per step, same results.
            results = {}
            # The batch whose Linears run LAST in the block's forward (known from the probe forward of the capture pass) is solved
            # on `self._late_stream`; the second forward below starts without it (see LATE_SOLVE).
            late = None
            if (LATE_SOLVE and propagate and self.dist_ctx is None and len(batches) > 1 and len(fired) == len(layers)
                    and set(fired) == set(layers) and torch.cuda.is_available()):
                tail = next(names for names in batches if fired[-1] in names)
                if set(tail) == set(fired[len(fired) - len(tail):]):
                    late = tail
            late_event = None
            main = torch.cuda.current_stream(self.device) if late is not None else None

            def run_solve(names):
                """The (N-stacked) solve of one batch -> (scale, zero, Q, codes, perm) over all of its rows."""
                sv = solvers[names[0]]
                cfg = sv.cfg
                if names is late:
                    # the float weights were allocated on the main stream and are DROPPED below (replaced by Q) while the late stream
                    # has not read them yet: tell the allocator, or the main stream recycles the block under the late solve's feet
                    # (seen as NaN Hessians one block later)
                    for n in names:
                        layers[n].weight.data.record_stream(torch.cuda.current_stream(self.device))
                if len(names) == 1:
                    W = layers[names[0]].weight.data
                else:
                    W = torch.cat([layers[n].weight.data for n in names], dim=0)
                scale, _, zp, Q = sv.fasterquant(
                    W, blocksize=cfg["block_size"], percdamp=cfg["percdamp"], groupsize=cfg["group_size"],
                    act_order=cfg["act_order"], hybrid_order=cfg["hybrid_order"], fp8_aware=cfg["fp8_aware"],
                    static_groups=cfg["static_groups"],
                )
                return scale, zp, Q, sv.codes, sv.export_perm

            def scatter(names, stacked):
                """Hand the rows of a solved batch back to its Linears."""
                scale, zp, Q, codes, perm = stacked
                cfg = solvers[names[0]].cfg
                r0 = 0
                for n in names:
                    rows = layers[n].weight.shape[0]
                    sl = slice(r0, r0 + rows)
                    r0 += rows
                    one = len(names) == 1
                    layers[n].weight.data = Q if one else Q[sl].contiguous()
                    results[n] = dict(scale=scale if one else scale[sl].contiguous(),
                                      zero=None if cfg["sym"] else (zp if one else zp[sl].contiguous()), perm=perm,
                                      codes=codes if one else codes[sl].contiguous())
                    solvers[n].perm = perm

            def solve(names):
                scatter(names, run_solve(names))

            with _phase("gptq.solve_issue"):
                if late is not None:
                    # issued FIRST (its factorisation is the longest chain), on its own stream; everything it allocates there and
                    # that the main stream reads later (the new weight, codes, scales) is handed over with record_stream
                    if self._late_stream is None:
                        self._late_stream = torch.cuda.Stream(device=self.device)
                    self._late_stream.wait_stream(main)
                    with torch.cuda.stream(self._late_stream):
                        solve(late)
                        late_event = self._late_stream.record_event()
                    for n in late:
                        for t in (layers[n].weight.data, results[n]["scale"], results[n]["zero"], results[n]["codes"], results[n]["perm"]):
                            if isinstance(t, torch.Tensor):
                                t.record_stream(main)
                if plan2d is not None:
                    self._solve_2d(plan2d, batches, solvers, layers, run_solve, scatter)
                else:
                    for names in batches:
                        if names is not late:
                            solve(names)

            def finish_solves():
                for acc, _, _, _ in distinct:
                    acc.check()  # deferred "not positive definite" checks of this group's factorisations (one sync each)
                for n in list(solvers):
                    solvers[n].free()

            if late is None:
                with _phase("gptq.solve_wait"):
                    finish_solves()
                del solvers, distinct
            # Step 2.5: outputs of the quantised block become the next block's inputs (reference :749-762)
            def replace(j, out):
                if "hidden_states" in self.cache_key_arguments:
                    self.cache_key_arguments["hidden_states"][j] = out
                else:
                    self.cache_positional_arguments[0][j] = out

            if propagate:
                gates = []
                if late is not None:
                    def gate(_, inp):  # the forward's stream waits for the late solve where it first needs its weight
                        torch.cuda.current_stream(self.device).wait_event(late_event)

                    gates = [layers[n].register_forward_pre_hook(gate) for n in late]
                try:
                    with _phase("gptq.second_forward"):
                        self._run_block(block, on_output=replace)
                finally:  # a forward that raises must not leave the gates on the user's modules
                    for h in gates:
                        h.remove()
            if late is not None:
                main.wait_event(late_event)  # (a block whose forward never reached those modules)
                with _phase("gptq.solve_wait"):
                    finish_solves()
                del solvers, distinct
            # Step 2.6: export to the packed module (reference :769-849) -- on device, from the emitted codes
            pack_phase = _phase("gptq.pack")
            pack_phase.__enter__()
            for name, layer in layers.items():
                cfg = self.get_layer_config(self.get_full_layer_name(name, block_idx))
                r = results[name]
                if isinstance(layer, nn.Linear):
                    in_features, out_features = layer.in_features, layer.out_features
                else:
                    in_features, out_features = layer.weight.shape[0], layer.weight.shape[1]
                new_module = MI355XWeightOnlyLinear(
                    in_features, out_features, dtype=cfg["dtype"], bits=cfg["bits"], group_size=cfg["group_size"],
                    zp=r["zero"] is not None, bias=layer.bias is not None, g_idx=r["perm"] is not None, device=self.device,
                )
                new_module.pack_codes(r["codes"], r["scale"], r["zero"], layer.bias, g_idx=r["perm"])
                set_module(block, name, new_module)
            pack_phase.__exit__()
            del results


class GPTQuantizer(INCQuantizer):
    """Algorithm plug-in (reference gptq.py:1651)."""

    def __init__(self, quant_config={}):
        super().__init__(quant_config)

    @torch.no_grad()
    def prepare(self, model, nsamples=128, max_seq_length=2048, use_max_length=True, device=None, use_layer_wise=False,
                model_path=None, quant_lm_head=False, use_block_wise=False, *args, **kwargs):
        assert isinstance(model, torch.nn.Module), "only support torch module"
        self.model_device = get_model_device(model)
        self.gptq_quantizer = RAWGPTQuantizer(
            model, weight_config=self.quant_config, nsamples=nsamples, use_max_length=use_max_length,
            max_seq_length=max_seq_length, device=device, use_layer_wise=use_layer_wise, model_path=model_path,
            quant_lm_head=quant_lm_head, use_block_wise=use_block_wise,
            **{k: v for k, v in kwargs.items() if k in ("share_hessians", "block_callback", "hessian_allreduce", "row_shard_solve", "independent_blocks", "solve_2d")},
        )
        self.gptq_quantizer.prepare_for_calibration()
        return self.gptq_quantizer.model

    @torch.no_grad()
    def convert(self, model, *args, **kwargs):
        self.gptq_quantizer.model = model
        self.gptq_quantizer.remove_prepare_for_calibration()
        q_model = self.gptq_quantizer.execute_quantization()
        logger.info("GPTQ quantizing done.")
        return q_model
