"""CPU oracle for the weight-only-quant hot path -- TEST INFRASTRUCTURE ONLY.

A restatement of the reference's algorithm (intel/neural-compressor v3.9, paths relative to
/root/reference/neural_compressor/torch/algorithms/weight_only/) in numpy (integer / byte work) and CPU torch
ops (floating point: the reference's own arithmetic library, incl. LAPACK potrf/potri through torch.linalg).
Nothing under neural_compressor_amd/ imports this module; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, as the checker / reported baseline -- never as the product path.

Pinning: every function here is checked against outputs of the unmodified reference (imported from
/root/reference in the build container) by tests/golden/make_golden.py -> tests/golden/*.npz, and the
fixtures are re-checked by tests/test_oracle_golden.py (runs without the reference and without a GPU).
"""

import math

import numpy as np
import torch

# =====================================================================================================
# bit packing (integer, bit-exact)
# =====================================================================================================
_NP_INT = {8: np.int8, 16: np.int16, 32: np.int32, 64: np.int64}


def pack_rows(raw, bits, cbits=32):
    """modules.py:528-544 (pack_tensor_with_numpy_impl) == numba packers bit_packer.py:35-278.

    raw [R,C] ints -> [R, ceil(C/n_pack)] words of `cbits` bits; element e of word j sits at bits [e*bits, (e+1)*bits).
    """
    raw = np.asarray(raw)
    n_pack = cbits // bits
    dt = _NP_INT[cbits]
    R, C = raw.shape
    out = np.zeros((R, math.ceil(C / n_pack)), dtype=dt)
    mask = np.uint8(2**bits - 1)
    for j in range(out.shape[1]):
        tmp = raw[:, n_pack * j : n_pack * (j + 1)].astype(dt)
        tmp &= mask
        for e in range(tmp.shape[1]):
            tmp[:, e] = np.left_shift(tmp[:, e], bits * e)
            out[:, j] |= tmp[:, e]
    return out


def unpack_rows(packed, bits, cbits=32, mask_sign=True):
    """modules.py:558-578 (unpack_tensor_with_numpy): arithmetic shifts on the signed container, `& mask` iff the
    module has qzeros; int16 out."""
    packed = np.asarray(packed)
    n_pack = cbits // bits
    out = np.zeros((packed.shape[0], packed.shape[1] * n_pack), dtype=np.int16)
    mask = np.uint8(2**bits - 1)
    for j in range(packed.shape[1]):
        for e in range(n_pack):
            tmp = packed[:, j]
            tmp = np.left_shift(tmp, cbits - bits * (e + 1))
            tmp = np.right_shift(tmp, cbits - bits)
            if mask_sign:
                tmp = tmp & mask
            out[:, j * n_pack + e] = tmp.astype(np.int16)
    return out


def woq_pack_optimum(int_weight, scales, zp, bits):
    """INCWeightOnlyLinear.pack, optimum format (modules.py:321-375).

    int_weight [N,K] ints (sym: signed, zp None; asym: 0..2^b-1 with zp [N,G]); scales [N,G] float.
    -> qweight [ceil(K/np), N] int32, qzeros [G, ceil(N/np)] int32, scales fp16 [G, N].
    """
    iw = np.asarray(int_weight).astype(np.int32).copy()
    scales = np.asarray(scales, dtype=np.float32)
    if zp is None:  # :329-334
        shift = 2 ** (bits - 1)
        iw = iw + shift
        z = np.zeros(scales.shape, dtype=np.int32) + shift
    else:
        z = np.asarray(zp).astype(np.int32).copy()
    qweight = pack_rows(iw, bits, 32)  # [N, K/np]   :357
    z = z - 1  # :364
    qzeros = pack_rows(z.T.copy(), bits, 32)  # [G, N/np]  :365-371
    return np.ascontiguousarray(qweight.T), qzeros, np.ascontiguousarray(scales.astype(np.float16).T)  # :372-375


def woq_unpack_optimum(qweight, qzeros, N, K, G, bits):
    """INCWeightOnlyLinear.unpack (modules.py:377-411) -> (int_weight [N,K] int16, zp [N,G] int16)."""
    w = unpack_rows(np.ascontiguousarray(np.asarray(qweight).T), bits, 32, True)[:N, :K]
    z = unpack_rows(np.asarray(qzeros), bits, 32, True)  # [G, N]
    z = np.ascontiguousarray(z.T)[:N, :G].astype(np.int16)
    z = z + 1  # :407-410
    z = np.where(z > (2**bits - 1), 0, z).astype(np.int16)
    return w, z


def woq_recover(qweight, scales_f16, qzeros, N, K, bits, group_size, g_idx=None):
    """INCWeightOnlyLinear.recover (modules.py:413-443) -> fp16 [N,K]: int8(w - zp[g]) * scales[g] in fp16."""
    G = scales_f16.shape[0]
    w, z = woq_unpack_optimum(qweight, qzeros, N, K, G, bits)
    s = np.ascontiguousarray(np.asarray(scales_f16).T)  # [N,G] fp16
    if g_idx is None:
        g_idx = np.arange(K) // group_size
    g_idx = np.asarray(g_idx).astype(np.int64)
    d = (w.astype(np.int16) - z[:, g_idx].astype(np.int16)).astype(np.int8)
    return (d.astype(np.float32) * s[:, g_idx].astype(np.float32)).astype(np.float16)


# =====================================================================================================
# group-wise RTN (float; torch CPU ops like the reference)
# =====================================================================================================
def qdq_weight_sym(weight, bits=4, quantile=1.0, return_int=False, full_range=False):
    """utility.py:199-244 (operates in place on `weight` [rows, group])."""
    maxq = torch.tensor(2 ** (bits - 1) - 1)
    minq = torch.tensor(-(2 ** (bits - 1)))
    if bits == 1:
        maxq = torch.tensor(2 ** (bits - 1))
        minq = torch.tensor(2 ** (bits - 1) - 1)
    max_val = torch.max(weight, 1)[0]
    min_val = torch.min(weight, 1)[0]
    flip_flag = torch.abs(max_val) > torch.abs(min_val)
    wmax = torch.max(torch.abs(max_val), torch.abs(min_val))
    wmax = wmax * quantile
    wmax[wmax == 0] = 1
    if full_range:
        scale = wmax / (-minq)
        scale = torch.where(flip_flag, -scale, scale)
    else:
        scale = wmax / maxq
    scale.unsqueeze_(dim=-1)
    weight.div_(scale)
    weight.round_()
    weight.clamp_(minq, maxq)
    if return_int:
        return weight, scale, None
    return weight.mul_(scale)


def qdq_weight_asym(weight, bits=4, quantile=1.0, return_int=False):
    """utility.py:162-196."""
    maxq = torch.tensor(2**bits - 1)
    zeros = torch.zeros(weight.shape[0])
    wmin = torch.minimum(weight.min(1)[0], zeros)
    wmax = torch.maximum(weight.max(1)[0], zeros)
    wmin = wmin * quantile
    wmax = wmax * quantile
    tmp = (wmin == 0) & (wmax == 0)
    wmin[tmp] = -1
    wmax[tmp] = +1
    scale = (wmax - wmin) / maxq
    zp = torch.round(-wmin / scale)
    scale.unsqueeze_(dim=-1)
    zp.unsqueeze_(dim=-1)
    weight.div_(scale)
    weight.round_()
    weight.add_(zp)
    weight.clamp_(0, maxq)
    if return_int:
        return weight, scale, zp
    weight.sub_(zp)
    return weight.mul_(scale)


# 4-bit float code books (utility.py:52-98) and the integers stored for their entries
NF4 = [-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
       -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
       0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0]
FP4_BNB = [-12.0, -8.0, -6.0, -4.0, -3.0, -2.0, -0.0625, 0, 0.0625, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0]
FP4_E2M1 = [-1.0, -0.6666666666666666, -0.5, -0.3333333333333333, -0.25, -0.16666666666666666, -0.010416666666666666, 0.0,
            0.010416666666666666, 0.16666666666666666, 0.25, 0.3333333333333333, 0.5, 0.6666666666666666, 1.0]
FLOAT_MAPPING = {"nf4": NF4, "fp4": FP4_BNB, "fp4_e2m1_bnb": FP4_BNB, "fp4_e2m1": FP4_E2M1}
INT_MAPPING = {"nf4": [7, 1, 2, 3, 4, 5, 6, 0, -8, -7, -6, -5, -4, -3, -2, -1], "fp4": [-5, -6, -3, -4, -1, -2, -7, 0, 1, 6, 7, 4, 5, 2, 3],
               "fp4_e2m1_bnb": [-5, -6, -3, -4, -1, -2, -7, 0, 1, 6, 7, 4, 5, 2, 3], "fp4_e2m1": [-1, -2, -3, -4, -5, -6, -7, 0, 1, 2, 3, 4, 5, 6, 7]}


def quantize_4bit(tensor, quantile=1.0, dtype="nf4", return_int=False, keep_scale=False, scale=None):
    """utility.py:112-149 (operates in place on `tensor` [rows, group]); keep_scale = the reference's double_quant kwarg,
    scale = its `scale` kwarg (:127-128: the caller's scale replaces the rows' own max)."""
    allow_data, allow_bit = FLOAT_MAPPING[dtype], INT_MAPPING[dtype]
    if scale is None:
        scale = tensor.abs().max(1)[0] * quantile / max(allow_data)
        scale.unsqueeze_(dim=-1)
    tensor.div_(scale)
    mid = [(allow_data[i] + allow_data[i + 1]) / 2 for i in range(len(allow_data) - 1)]
    q = torch.zeros_like(tensor)
    for i in range(len(allow_data)):
        data = allow_bit[i] if return_int else allow_data[i]
        if i == 0:
            q += torch.where(tensor <= mid[i], data, 0)
        elif i == len(allow_data) - 1:
            q += torch.where(tensor > mid[i - 1], data, 0)
        else:
            q += torch.where((mid[i - 1] < tensor) & (tensor <= mid[i]), data, 0)
    tensor.copy_(q)
    if return_int or keep_scale:
        return tensor, scale, None
    return tensor.mul_(scale)


def quant_tensor(weight, bits=4, group_size=-1, scheme="asym", quantile=1.0, return_int=False, full_range=False, dtype="int",
                 double_quant=False, double_quant_dtype="int", double_quant_bits=8, double_quant_scheme="asym",
                 double_quant_group_size=256, double_quant_return_int=False):
    """utility.py:272-436 for dtype "int" / "nf4" / "fp4*", optionally with double quantisation of the scales (:378-436).
    NOT in place (works on a clone).

    Returns the fake-quantised weight, or (int_weight, scale [N,G], zp [N,G] | None) when return_int.
    """
    if double_quant:
        w, scale, zp = quant_tensor(weight, bits, group_size, scheme, quantile, True, full_range, dtype)
        shape = scale.shape
        flat = scale.reshape(1, -1).clone()
        mean = None
        if double_quant_scheme == "asym":
            mean = flat.mean()
            flat.sub_(mean)
        if double_quant_return_int:  # :383-384, 393-405: the inner call's result is dropped, so the unpack of `scale` fails -- always
            raise ValueError("not enough values to unpack (expected 3, got 1)")
        flat = quant_tensor(flat, double_quant_bits, double_quant_group_size, "sym", 1.0, False, False, double_quant_dtype)
        if mean is not None:
            flat.add_(mean)
        scale = flat.reshape(shape)
        if return_int:
            return w, scale, zp
        N, K = weight.shape
        gs = K if (group_size == -1 or K < group_size) else group_size
        vals = w.clone()
        if dtype in FLOAT_MAPPING:  # without return_int the reference's actor leaves the code-book VALUES in the tensor (:134)
            lut = torch.zeros(16)
            lut[torch.tensor(INT_MAPPING[dtype]) + 8] = torch.tensor(FLOAT_MAPPING[dtype], dtype=torch.float32)
            vals = lut[(w + 8).long()]
        if zp is not None:
            vals = vals - zp.repeat_interleave(gs, 1)[:, :K]
        return vals * scale.repeat_interleave(gs, 1)[:, :K]
    weight = weight.clone()

    def actor(w):
        if dtype in FLOAT_MAPPING:
            return quantize_4bit(w, quantile, dtype, return_int)
        if scheme == "sym":
            return qdq_weight_sym(w, bits, quantile, return_int, full_range)
        return qdq_weight_asym(w, bits, quantile, return_int)

    if group_size == -1 or weight.shape[1] < group_size:
        group_size = weight.shape[1]
    orig_shape = weight.shape
    if weight.shape[1] % group_size == 0:
        res = actor(weight.reshape(-1, group_size))
        if return_int:
            w, scale, zp = res
            return (w.reshape(orig_shape), scale.reshape(orig_shape[0], -1), None if zp is None else zp.reshape(orig_shape[0], -1))
        return res.reshape(orig_shape)
    split = weight.shape[1] // group_size * group_size  # :334-376 tail group
    r1 = actor(weight[:, :split].reshape(-1, group_size))
    r2 = actor(weight[:, split:].clone())
    if return_int:
        w1, s1, z1 = r1
        w2, s2, z2 = r2
        w = torch.cat([w1.reshape(orig_shape[0], split), w2], dim=1)
        scale = torch.cat([s1.reshape(orig_shape[0], -1), s2], dim=1)
        zp = None if z1 is None else torch.cat([z1.reshape(orig_shape[0], -1), z2], dim=1)
        return w, scale, zp
    return torch.cat([r1.reshape(orig_shape[0], split), r2], dim=1)


def search_clip(weight, bits=4, group_size=32, scheme="asym", full_range=False):
    """utility.py:439-480 on a weight tensor."""
    best_error, best_ratio = float("inf"), None
    n_grid, max_shrink = 200, 0.2
    for i_s in range(int(max_shrink * n_grid)):
        ratio = 1 - i_s / n_grid
        q = quant_tensor(weight, bits=bits, group_size=group_size, scheme=scheme, full_range=full_range, quantile=ratio)
        loss = (weight - q).float().pow(2).mean()
        if loss < best_error:
            best_error, best_ratio = loss, ratio
    return best_ratio


# =====================================================================================================
# GPTQ
# =====================================================================================================
class GPTQQuantParams:
    """Quantizer.find_params / quantize for dtype int, perchannel, weight=True (gptq.py:1501-1637), incl. the `mse`
    shrink-grid search (:1567-1584; configure() defaults norm=2.4, grid=100, maxshrink=0.8, :1375-1387)."""

    def __init__(self, bits=4, sym=False, mse=False, norm=2.4, grid=100, maxshrink=0.8):
        self.bits, self.sym = bits, sym
        self.mse, self.norm, self.grid, self.maxshrink = mse, norm, grid, maxshrink
        self.maxq = 2**bits - 1
        self.scale = torch.zeros(1)
        self.zero = torch.zeros(1)

    def find_params(self, x):
        x = x.flatten(1)
        tmp = torch.zeros(x.shape[0])
        xmin = torch.minimum(x.min(1)[0], tmp)
        xmax = torch.maximum(x.max(1)[0], tmp)
        if self.sym:
            xmax = torch.maximum(torch.abs(xmin), xmax)
            neg = xmin < 0
            if torch.any(neg):
                xmin[neg] = -xmax[neg]
        z = (xmin == 0) & (xmax == 0)
        xmin[z] = -1
        xmax[z] = +1
        self.scale = (xmax - xmin) / self.maxq
        if self.sym:
            self.zero = torch.full_like(self.scale, (self.maxq + 1) / 2)
        else:
            self.zero = torch.round(-xmin / self.scale)
        if self.mse:
            best = torch.full([x.shape[0]], float("inf"))
            for i in range(int(self.maxshrink * self.grid)):
                p = 1 - i / self.grid
                xmin1 = p * xmin
                xmax1 = p * xmax
                scale1 = (xmax1 - xmin1) / self.maxq
                zero1 = torch.round(-xmin1 / scale1) if not self.sym else self.zero
                q = torch.clamp(torch.round(x / scale1.unsqueeze(1)) + zero1.unsqueeze(1), 0, self.maxq)
                q = scale1.unsqueeze(1) * (q - zero1.unsqueeze(1))
                q -= x
                q.abs_()
                q.pow_(self.norm)
                err = torch.sum(q, 1)
                better = err < best
                if torch.any(better):
                    best[better] = err[better]
                    self.scale[better] = scale1[better]
                    self.zero[better] = zero1[better]
        self.scale = self.scale.reshape(-1, 1)
        self.zero = self.zero.reshape(-1, 1)

    def quantize(self, x):
        q = torch.clamp(torch.round(x / self.scale) + self.zero, 0, self.maxq)
        return self.scale * (q - self.zero)

    def ready(self):
        return torch.all(self.scale != 0)


def gptq_add_batch(H, nsamples, inp):
    """GPTQ.add_batch (gptq.py:1111-1141); returns (H, nsamples)."""
    if len(inp.shape) == 2:
        inp = inp.unsqueeze(0)
    tmp = inp.shape[0]
    if len(inp.shape) == 3:
        inp = inp.reshape((-1, inp.shape[-1]))
    inp = inp.t()
    H = H * (nsamples / (nsamples + tmp))
    nsamples += tmp
    inp = math.sqrt(2 / nsamples) * inp.float()
    H = H + inp.matmul(inp.t())
    return H, nsamples


def gptq_damped(H, percdamp=0.01):
    """gptq.py:1186-1189, 1221-1227: dead-column fix + damping -> (damped copy of H, dead mask)."""
    H = H.clone()
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    damp = percdamp * torch.mean(torch.diag(H))
    diag = torch.arange(H.shape[0])
    H[diag, diag] += damp
    return H, dead


def gptq_hinv(H, percdamp=0.01):
    """gptq.py:1186-1189, 1221-1231 -> (Hinv upper Cholesky factor of the damped inverse, dead mask). H is copied."""
    H, dead = gptq_damped(H, percdamp)
    H = torch.linalg.cholesky(H)
    H = torch.cholesky_inverse(H)
    H = torch.linalg.cholesky(H, upper=True)
    return H, dead


def gptq_hybrid_perms(diag_H, groupsize):
    """Quantizer.compute_local_perms / compute_global_perm / compose_final_perm (gptq.py:1389-1461): inside every group the columns
    in descending order of diag(H), the groups in descending order of their largest diag(H).  Returns (final_perm [K], global_perm [G])."""
    n = diag_H.numel()
    num_groups = n // groupsize
    local_perms, metric = [], []
    for g in range(num_groups):
        sub = diag_H[g * groupsize : (g + 1) * groupsize]
        local_perms.append(torch.argsort(sub, descending=True))
        metric.append(sub.max().item())
    global_perm = torch.argsort(torch.tensor(metric), descending=True)
    final = []
    for new_group in range(num_groups):
        orig = global_perm[new_group].item()
        for idx in local_perms[orig]:
            final.append(idx.item() + orig * groupsize)
    return torch.tensor(final, dtype=torch.long), global_perm


def gptq_fasterquant(W, H, bits=4, sym=False, blocksize=128, percdamp=0.01, groupsize=-1, act_order=False,
                     static_groups=False, Hinv=None, mse=False, trace=False, hybrid_order=False):
    """GPTQ.fasterquant (gptq.py:1143-1351) for int dtype.  Returns dict(scale [N,G], zero [N,G], Q fp32 [N,K], perm).

    `Hinv` (optional) injects a precomputed factor so the column loop can be tested in isolation.
    `trace` (test instrumentation, not part of the reference): also return `Win` [N,K], the value of every weight at
    the moment it was quantised (loop order, i.e. permuted columns with act_order) -- what a parity test needs to tell
    a rounding-tie flip (|w/scale| within float noise of a .5 boundary) from a real difference.
    """
    W = W.clone().float()
    quantizer = GPTQQuantParams(bits, sym, mse=mse)
    columns = W.shape[1]
    if not quantizer.ready():
        quantizer.find_params(W)
    H = H.clone()
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    groups = []
    if static_groups:
        import copy

        for i in range(0, columns, groupsize):
            q = copy.deepcopy(quantizer)
            q.find_params(W[:, i : i + groupsize])
            groups.append(q)
    perm = None
    final_perm = global_perm = None
    if hybrid_order:  # gptq.py:1203-1209: columns rearranged by diag(H) WITHOUT mixing groups
        assert not act_order, "Error: hybrid_act_order is not allowed with act_order"
        final_perm, global_perm = gptq_hybrid_perms(torch.diag(H), groupsize)
        W = W[:, final_perm]
        H = H[final_perm][:, final_perm]
    if act_order:
        perm = torch.argsort(torch.diag(H), descending=True)
        W = W[:, perm]
        H = H[perm][:, perm]
    Q = torch.zeros_like(W)
    if Hinv is None:
        damp = percdamp * torch.mean(torch.diag(H))
        diag = torch.arange(columns)
        H[diag, diag] += damp
        H = torch.linalg.cholesky(H)
        H = torch.cholesky_inverse(H)
        Hinv = torch.linalg.cholesky(H, upper=True)
    scale, zero = [], []
    Win = torch.zeros_like(W) if trace else None
    for i1 in range(0, columns, blocksize):
        i2 = min(i1 + blocksize, columns)
        count = i2 - i1
        W1 = W[:, i1:i2].clone()
        Q1 = torch.zeros_like(W1)
        Err1 = torch.zeros_like(W1)
        Hinv1 = Hinv[i1:i2, i1:i2]
        for i in range(count):
            w = W1[:, i]
            d = Hinv1[i, i]
            if groupsize != -1:
                if not static_groups:
                    if (i1 + i) % groupsize == 0:
                        quantizer.find_params(W[:, (i1 + i) : (i1 + i + groupsize)])
                        scale.append(quantizer.scale)
                        zero.append(quantizer.zero)
                else:
                    idx = i1 + i
                    if act_order:
                        idx = perm[idx]
                    quantizer = groups[idx // groupsize]
            if trace:
                Win[:, i1 + i] = w
            q = quantizer.quantize(w.unsqueeze(1)).flatten()
            Q1[:, i] = q
            err1 = (w - q) / d
            W1[:, i:] -= err1.unsqueeze(1).matmul(Hinv1[i, i:].unsqueeze(0))
            Err1[:, i] = err1
        Q[:, i1:i2] = Q1
        W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])
    if hybrid_order:  # gptq.py:1320-1328: columns back in place, the groups' parameters back in the groups' original order
        inv_final = torch.empty_like(final_perm)
        inv_final[final_perm] = torch.arange(final_perm.numel())
        Q = Q[:, inv_final]
        inv_global = torch.empty_like(global_perm)
        inv_global[global_perm] = torch.arange(global_perm.numel())
        scale = [scale[i] for i in inv_global.tolist()]
        zero = [zero[i] for i in inv_global.tolist()]
    if act_order:
        Q = Q[:, torch.argsort(perm)]
    # NB with static_groups the reference never appends to `scale` (gptq.py:1273-1277), so it returns only the
    # LAST group's parameters (:1341-1345) -- restated as is; only Q is meaningful in that mode.
    if scale == []:
        scale.append(quantizer.scale)
        zero.append(quantizer.zero)
    out = dict(scale=torch.cat(scale, dim=1), zero=torch.cat(zero, dim=1), Q=Q, perm=perm, final_perm=final_perm)
    if trace:
        out["Win"] = Win
    return out


def quant_weight_w_scale(weight, scale, zp=None, group_size=-1):
    """utility.py:483-537 for int dtype: ints = round(W/scale (+zp)) per group (NOT in place)."""
    weight = weight.clone()
    if group_size == -1:
        return weight.div_(scale).round_() if zp is None else weight.div_(scale).add_(zp).round_()
    int_weight = torch.zeros(weight.shape)
    leng = weight.shape[1] // group_size
    tail = weight.shape[1] % group_size != 0
    for i in range(leng):
        t = weight[:, i * group_size : (i + 1) * group_size].div_(scale[:, i].unsqueeze(1))
        if zp is not None:
            t.add_(zp[:, i].unsqueeze(1))
        int_weight[:, i * group_size : (i + 1) * group_size].copy_(t.round_())
    if tail:
        t = weight[:, leng * group_size :].div_(scale[:, -1].unsqueeze(1))
        if zp is not None:
            t.add_(zp[:, -1].unsqueeze(1))
        int_weight[:, leng * group_size :].copy_(t.round_())
    return int_weight


def gptq_export_ints(Q, scale, zero, sym, group_size, perm=None):
    """The export step of RAWGPTQuantizer.execute_quantization (gptq.py:795-813): int32 weights for pack()."""
    Q = Q.clone()
    if perm is not None:
        Q = Q[:, perm]
    ints = quant_weight_w_scale(Q, scale, None if sym else zero, group_size)
    if perm is not None:
        ints = ints[:, torch.argsort(perm)]
    return ints.type(torch.int32)


# =====================================================================================================
# AutoAWQ checkpoint repack
# =====================================================================================================
AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]  # utility.py:1257


def awq_unpack_fields(words, bits=4):
    """unpack_awq's integer part (utility.py:1305-1327): [R, C/8] int32 -> [R, C] with the AWQ field order undone
    (awq_reverse_reorder_int_tensor :1246-1270: column 8c + AWQ_ORDER[i] is field i)."""
    w = np.asarray(words).astype(np.uint32)
    n_pack = 32 // bits
    raw = np.stack([(w >> np.uint32(bits * i)) & np.uint32(2**bits - 1) for i in range(n_pack)], axis=-1)  # [R, C/8, 8]
    out = np.empty_like(raw)
    for i, col in enumerate(AWQ_ORDER):
        out[..., col] = raw[..., i]
    return out.reshape(w.shape[0], -1).astype(np.int32)


def awq_repack_to_optimum(awq_qweight, awq_qzeros, bits=4):
    """repack_awq_to_optimum_format (utility.py:1426-1459), integer restatement: the reference dequantises to fp16
    (unpack_awq :1329-1341) and re-rounds (pack_from_tensors :1395-1399), which returns the same integers.
    -> qweight [K/8, N] int32 (K-major fields), qzeros [G, N/8] int32 (sequential fields of zero-1)."""
    codes = awq_unpack_fields(awq_qweight, bits)  # [K, N]
    zeros = awq_unpack_fields(awq_qzeros, bits)   # [G, N]
    qweight = pack_rows(np.ascontiguousarray(codes.T), bits, 32)  # [N, K/8]  (:1404-1413)
    qzeros = pack_rows((zeros - 1) & (2**bits - 1), bits, 32)      # [G, N/8]  (:1415-1431)
    return np.ascontiguousarray(qweight.T), qzeros


# =====================================================================================================
# AWQ statistics
# =====================================================================================================
def awq_weight_scale(weight, q_group_size=-1):
    """awq.py:131-147."""
    org_shape = weight.shape
    if q_group_size > 0:
        weight = weight.view(-1, q_group_size)
    scale = weight.abs() / weight.abs().amax(dim=1, keepdim=True)
    return scale.view(org_shape).mean(0)


def awq_act_scale(input_val):
    """awq.py:151-154."""
    tmp = torch.cat([x.abs().view(-1, x.shape[-1]) for x in input_val], dim=0)
    return tmp.mean(0)


def awq_search_scale_module(weight, bias, input_val, group_size=32, scheme="asym", full_range=False):
    """ActAwareWeightQuant.search_scale (awq.py:264-361) for a module tuple of ONE Linear (the `module_inference`
    branch :311-313, :340-342): the 20-point alpha grid.  `input_val`: list of [.., K] activations of that Linear.
    Returns dict(history [20 python floats], best_index, best_scales [K]).  The search quantises as 4-bit integers
    whatever the configured width (the reference passes `num_bits=` / `data_type=`, which quant_tensor swallows)."""
    w_max = awq_weight_scale(weight, q_group_size=group_size)
    x_max = awq_act_scale(input_val)
    org_out = [torch.nn.functional.linear(x, weight, bias) for x in input_val]
    best_error, best_scales, best_i, history = float("inf"), None, None, []
    n_grid = 20
    for step in range(n_grid):
        ratio = step * 1 / n_grid
        scales = (x_max.pow(ratio) / w_max.pow(1 - ratio)).clamp(min=1e-4).view(-1)
        scales = scales / (scales.max() * scales.min()).sqrt()
        wq = quant_tensor(weight.mul(scales.view(1, -1)), bits=4, group_size=group_size, scheme=scheme, full_range=full_range) / scales.view(1, -1)
        loss = 0
        for x, o1 in zip(input_val, org_out):
            o2 = torch.nn.functional.linear(x, wq, bias)
            loss += (o1 - o2).float().pow(2).mean().item()
        history.append(loss)
        if loss < best_error:
            best_error, best_scales, best_i = loss, scales, step
    return dict(history=history, best_index=best_i, best_scales=best_scales.view(-1))


def awq_search_clip_module(weight, bias, input_val, group_size=32, scheme="asym", full_range=False, input_scale=None):
    """ActAwareWeightQuant.search_clip (awq.py:393-470) for one module: 10 clip ratios 1.00 .. 0.91, output-MSE criterion.
    `input_scale` [K]: the module is a MulLinear (modules.py:907-949: forward = linear(x * input_scale))."""
    xs = input_val if input_scale is None else [x * input_scale for x in input_val]
    org_out = [torch.nn.functional.linear(x, weight, bias) for x in xs]
    best_error, best_ratio, best_i, history = float("inf"), None, None, []
    n_grid, max_shrink = 100, 0.1
    for i_s in range(int(max_shrink * n_grid)):
        ratio = 1 - i_s / n_grid
        wq = quant_tensor(weight, bits=4, group_size=group_size, scheme=scheme, full_range=full_range, quantile=ratio)
        loss = 0
        for x, o1 in zip(xs, org_out):
            o2 = torch.nn.functional.linear(x, wq, bias)
            loss += (o1 - o2).float().pow(2).mean().item()
        history.append(loss)
        if loss < best_error:
            best_error, best_ratio, best_i = loss, ratio, i_s
    return dict(history=history, best_index=best_i, best_ratio=best_ratio)


# =====================================================================================================
# forward
# =====================================================================================================
def woq_dense_weight(qweight, scales_f16, qzeros, N, K, bits, group_size, compute_dtype=torch.bfloat16, g_idx=None):
    """The dense weight forward() multiplies by: recover() (modules.py:413-443) is the exact product
    int8(q - zp) * scale rounded once to fp16; the kernels round the same exact product to `compute_dtype` instead."""
    G = scales_f16.shape[0]
    iw, z = woq_unpack_optimum(qweight, qzeros, N, K, G, bits)
    gi = (np.arange(K) // group_size) if g_idx is None else np.asarray(g_idx)
    d = (iw.astype(np.int16) - z[:, gi].astype(np.int16)).astype(np.int8).astype(np.float32)
    s = np.ascontiguousarray(np.asarray(scales_f16).T).astype(np.float32)[:, gi]
    return torch.from_numpy(d * s).to(compute_dtype).float()


def woq_linear(x, qweight, scales_f16, qzeros, bias, N, K, bits, group_size, compute_dtype=torch.bfloat16, g_idx=None, dense=None):
    """INCWeightOnlyLinear.forward (modules.py:594-610) == F.linear(x, recover(), bias), evaluated in fp32 on the
    weight rounded to `compute_dtype` (SURVEY.md 8(c) comparator (4)).  `dense`: a cached woq_dense_weight(...) of the
    same arguments (tests that call this many times on one layer)."""
    w = dense if dense is not None else woq_dense_weight(qweight, scales_f16, qzeros, N, K, bits, group_size, compute_dtype, g_idx)
    y = torch.nn.functional.linear(x.to(compute_dtype).float(), w, None if bias is None else bias.float())
    return y


# =====================================================================================================
# SmoothQuant W8A8 (BASELINE config #4): neural_compressor/torch/algorithms/smooth_quant/utility.py
# =====================================================================================================
def sq_cal_scale(input_max_abs, weights, alpha, weight_max_lb=1e-5):
    """cal_scale (:605-626): `weights` = list of [N_i, K] tensors that share the input."""
    weights = torch.cat(weights, dim=0)
    weight_max = torch.max(torch.abs(weights), dim=0)[0]
    weight_max = torch.clip(weight_max, weight_max_lb)
    input_power = torch.pow(input_max_abs, alpha)
    weight_power = torch.pow(weight_max, 1 - alpha)
    weight_scale = torch.clip(input_power / weight_power, min=1e-5)
    weight_scale[input_power == 0] = 1.0
    return weight_scale


def sq_quant_w(w, num_bits=8):
    """quant_dequant_w_v1 (:652-695), Linear, scheme "sym" -> (codes int32 [N,K], scale fp32 [N], dequantised [N,K])."""
    eps = torch.finfo(torch.float32).eps
    q_min, q_max = -(2.0 ** (num_bits - 1)), 2.0 ** (num_bits - 1) - 1.0
    x_max = torch.max(torch.abs(w), dim=1).values
    scale = x_max / (float(q_max - q_min) / 2)
    scale = torch.clip(scale, min=eps).unsqueeze(dim=-1)
    q = torch.round(w / scale)
    q.clamp_(q_min, q_max)
    return q.to(torch.int32), scale.squeeze(-1), q * scale


def sq_quant_w_asym(w, num_bits=8):
    """quant_dequant_w_v1 (:652-695), Linear, scheme "asym": per-output-channel uint8 with a zero point -> dequantised [N,K].
    NB the range includes 0 (x_max / x_min are clamped against a zero vector) but the zero point is round(-min(row) / scale) with
    the UNclamped row minimum (:689), as the reference writes it."""
    eps = torch.finfo(torch.float32).eps
    q_min, q_max = 0, 2.0**num_bits - 1.0
    tmp = torch.zeros(w.shape[0])
    x_max = torch.maximum(torch.max(w, dim=1).values, tmp)
    x_min = torch.minimum(torch.min(w, dim=1).values, tmp)
    scale = torch.clip((x_max - x_min) / (2**num_bits - 1), min=eps)
    bias = torch.round(0 - (torch.min(w, dim=1).values) / scale).unsqueeze(dim=-1)
    scale = scale.unsqueeze(dim=-1)
    q = torch.round(w / scale + bias)
    q.clamp_(q_min, q_max)
    return (q - bias) * scale


def sq_act_qparams(input_scale, input_min, input_max):
    """SQLinearWrapper._calculate_qparams (:2607-2631) for torch.quint8 -> (scale, zero_point) python floats."""
    min_val = torch.min(input_min * input_scale)
    max_val = torch.max(input_max * input_scale)
    min_val_neg = torch.min(min_val, torch.zeros_like(min_val))
    max_val_pos = torch.max(max_val, torch.zeros_like(max_val))
    scale = (max_val_pos - min_val_neg) / 255.0
    scale = torch.max(scale, torch.tensor(torch.finfo(torch.float32).eps))
    zero_point = 0 - torch.round(min_val_neg / scale).to(torch.int)
    zero_point = torch.clamp(zero_point, 0, 255)
    return float(scale), int(zero_point)


def sq_quant_x(x, scale, zero_point):
    """The integer half of quant_dequant_x_v1 (:726-755) with given static parameters -> uint8 codes as int32."""
    q = torch.round(x / scale + float(zero_point))
    q.clamp_(0, 255)
    return q.to(torch.int32)


def sq_quant_dequant_x(x, min_x, max_x, num_bits=8):
    """quant_dequant_x_v1 (:726-755) as written (parameters from the tensor-wide min / max)."""
    eps = torch.finfo(torch.float32).eps
    if max_x is None or min_x is None:  # dynamic form (:745-746)
        max_x, min_x = torch.max(x), torch.min(x)
    else:
        max_x, min_x = torch.max(max_x), torch.min(min_x)
    scale = (max_x - min_x) / (2**num_bits - 1)
    scale = torch.clip(scale, min=eps)
    bias = torch.round((0 - min_x) / scale)
    q_x = torch.round(x / scale + bias)
    q_x.clamp_(0, 2.0**num_bits - 1.0)
    return scale * (q_x - bias)


def sq_w8a8_linear(x, w_smoothed, input_scale, act_scale, act_zp, bias=None):
    """What the W8A8 module computes, in integers then one fp32 scaling:
    y = s_x * s_w[n] * sum_k (qx - zp) * qw + b   with qx = quant(x * input_scale), qw = quant_w(w_smoothed)."""
    xs = x if input_scale is None else x * input_scale
    qx = sq_quant_x(xs.float(), act_scale, act_zp)
    qw, sw, _ = sq_quant_w(w_smoothed.float())
    acc = (qx.to(torch.int64) - act_zp) @ qw.to(torch.int64).T
    y = acc.to(torch.float64) * (act_scale * sw.to(torch.float64)).unsqueeze(0)
    if bias is not None:
        y = y + bias.to(torch.float64)
    return y.to(torch.float32)
