"""A/B partners of product kernels -- TEST INFRASTRUCTURE, never imported by the package.

Round 4's product carried these behind environment switches; they live here so that the product has exactly one path:
  * inverse_cholesky_upper_python: the blocked inverse-Cholesky factor driven from Python with torch.mm for every product and
    inc_chol_diag_block for the 128 x 128 leaves (rounds 1-3) -- the partner of inc_gptq_inverse_factor, and the form whose HOST
    LOGIC tests/test_cholesky_host_logic.py checks on the CPU with a torch.linalg stand-in for the leaf kernel;
  * exact_trio: the reference's three factorisations (gptq.py:1228-1230) through torch.linalg (rocSOLVER on the GPU);
  * python_column_loop: the GPTQ column loop (gptq.py:1250-1304) as one launch per piece from Python -- the partner of
    inc_gptq_quantize_layer, with or without the look-ahead stream.
"""

import torch

from neural_compressor_amd import ops

CHOL_NB = 128
QBLOCK = 128


@torch.no_grad()
def exact_trio(H):
    L = torch.linalg.cholesky(H)
    Hi = torch.cholesky_inverse(L)
    return torch.linalg.cholesky(Hi, upper=True).contiguous()


@torch.no_grad()
def inverse_cholesky_upper_python(H, check=True, outer=1024, tri_depth=2, tri_min=512, diag_block=None, side=None):
    """U = upper Cholesky factor of H^-1 as J Lr^-1 J (see gptq.inverse_cholesky_upper), every O(K^3) product a torch.mm.
    `diag_block(A_view, Linv_view, info, tag)`: the leaf kernel (default ops.chol_diag_block); `side`: a second stream for the
    look-ahead over outer blocks (same GEMMs, same operands: same bits)."""
    CHOL_OUTER, TRI_DEPTH, TRI_MIN = outer, tri_depth, tri_min
    diag_block = diag_block or ops.chol_diag_block
    assert H.dim() == 2 and H.shape[0] == H.shape[1] and H.dtype == torch.float32
    K = H.shape[0]
    nb = CHOL_NB
    Kp = -(-K // nb) * nb
    dev = H.device
    if Kp == K:
        A = torch.flip(H, (0, 1)).contiguous()
    else:  # pad with an identity block: chol(blockdiag(Hr, I)) = blockdiag(Lr, I)
        A = torch.zeros((Kp, Kp), dtype=torch.float32, device=dev)
        A[:K, :K] = torch.flip(H, (0, 1))
        A.diagonal()[K:] = 1.0
    X = torch.zeros((Kp, Kp), dtype=torch.float32, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)

    def mm_tri_right(C, T, out, depth):
        # out = C @ T for a LOWER-triangular T without multiplying its zero half: T = [[a, 0], [b, c]] ->
        # [C1 a + C2 b, C2 c], a and c recursively (each level drops a quarter of the remaining flops)
        n = T.shape[0]
        h = (n // 2 // nb) * nb
        if depth == 0 or h < TRI_MIN or h == 0:
            torch.mm(C, T, out=out)
            return
        mm_tri_right(C[:, :h], T[:h, :h], out[:, :h], depth - 1)
        out[:, :h].addmm_(C[:, h:], T[h:, :h])
        mm_tri_right(C[:, h:], T[h:, h:], out[:, h:], depth - 1)

    def mm_tri_left(T, B, out, depth, alpha=1.0):
        # out = alpha * T @ B for a LOWER-triangular T: [[a, 0], [b, c]] @ [B1; B2] = [a B1; b B1 + c B2]
        n = T.shape[0]
        h = (n // 2 // nb) * nb
        if depth == 0 or h < TRI_MIN or h == 0:
            torch.mm(T, B, out=out)
            if alpha != 1.0:
                out.mul_(alpha)
            return
        mm_tri_left(T[:h, :h], B[:h], out[:h], depth - 1, alpha)
        mm_tri_left(T[h:, h:], B[h:], out[h:], depth - 1, alpha)
        out[h:].addmm_(T[h:, :h], B[:h], alpha=alpha)

    def invert_by_doubling(segs):
        # Lr^-1 of the span covered by `segs` = [(start, size), ...], whose diagonal blocks of X already hold the inverses.
        # Both factors of  X21 = -X22 (C X11)  are lower-triangular inverses: the products skip their zero halves (two levels
        # of 2 x 2 splitting: 62 % of the flops of a full GEMM; more than half of the factorisation's flops are in here).
        while len(segs) > 1:
            nxt = []
            for p in range(0, len(segs) - 1, 2):
                (s1, n1), (s2, n2) = segs[p], segs[p + 1]
                C = A[s2:s2 + n2, s1:s1 + n1]
                T = torch.empty((n2, n1), dtype=torch.float32, device=dev)
                mm_tri_right(C, X[s1:s1 + n1, s1:s1 + n1], T, TRI_DEPTH)
                mm_tri_left(X[s2:s2 + n2, s2:s2 + n2], T, X[s2:s2 + n2, s1:s1 + n1], TRI_DEPTH, alpha=-1.0)
                nxt.append((s1, n1 + n2))
            if len(segs) % 2:
                nxt.append(segs[-1])
            segs = nxt
        return segs[0]

    # Two-level blocking (only the LOWER triangle of A is read or kept up to date): an outer block of CHOL_OUTER columns is
    # factored with 128-wide steps confined to its own diagonal block, its factor is inverted by doubling, and then ONE panel
    # solve and ONE trailing update of depth CHOL_OUTER serve the rest of the matrix -- 11 deep GEMMs at K = 11008 instead of 86
    # rank-128 updates of the whole trailing matrix (which ran at a third of the library's fp32 GEMM rate and made the K = 11008
    # factorisation the critical path of a block: 46 ms, profiles/r2d).
    outer = max(nb, (CHOL_OUTER // nb) * nb)
    tag = 0
    top = []
    # Look-ahead over the outer blocks (gptq.CHOL_LOOKAHEAD = True): the next outer block needs only the FIRST column
    # chunk of this block's trailing update (it holds that block's diagonal block and its whole panel).  The other chunks run on
    # a second stream underneath the next block's factorisation -- a chain of one-workgroup diagonal kernels and small GEMMs that
    # leaves the chip idle (kernel trace at K = 11008: 9.9 ms of chol_diag_block + 13.7 ms of GEMMs back to back) -- and are
    # awaited before the next trailing update, which accumulates into the same columns.  Same GEMMs, same operands: same bits.
    main = torch.cuda.current_stream(dev) if H.is_cuda else None
    side = side if (side is not None and H.is_cuda and Kp > 2 * outer) else None
    pending = None  # event: the remaining chunks of the previous trailing update are done
    keep = []       # operands still read by the side stream
    for B in range(0, Kp, outer):
        n2 = min(outer, Kp - B)
        D = A[B:B + n2, B:B + n2]
        XD = X[B:B + n2, B:B + n2]
        for j in range(0, n2, nb):
            tag += 1
            diag_block(D[j:j + nb, j:j + nb], XD[j:j + nb, j:j + nb], info, tag)
            if j + nb < n2:
                panel = D[j + nb:, j:j + nb]                       # [m, nb] strided view
                lp = torch.mm(panel, XD[j:j + nb, j:j + nb].t())   # L_panel = A_panel @ inv(L_jj)^T
                panel.copy_(lp)
                D[j + nb:, j + nb:].addmm_(lp, lp.t(), alpha=-1.0)
        top.append(invert_by_doubling([(B + j, nb) for j in range(0, n2, nb)]))
        if B + n2 < Kp:
            panel = A[B + n2:, B:B + n2]                            # [M, n2]
            lp = torch.mm(panel, XD.t())                            # L_panel = A_panel @ inv(L_DD)^T  (XD^T upper-triangular)
            panel.copy_(lp)
            M = Kp - (B + n2)
            chunk = max(outer, -(-M // 6 // nb) * nb)                # lower triangle only: <= 6 column chunks, each from its diagonal down
            if pending is not None:
                main.wait_event(pending)                            # the previous update's remaining chunks wrote these columns
                pending = None
            first = True
            for c0 in range(0, M, chunk):
                c1 = min(c0 + chunk, M)
                if side is not None and not first:
                    if c0 == chunk:
                        ready = torch.cuda.Event()
                        ready.record(main)                          # lp and the first chunk are complete
                        side.wait_event(ready)
                    with torch.cuda.stream(side):
                        A[B + n2 + c0:, B + n2 + c0:B + n2 + c1].addmm_(lp[c0:], lp[c0:c1].t(), alpha=-1.0)
                else:
                    A[B + n2 + c0:, B + n2 + c0:B + n2 + c1].addmm_(lp[c0:], lp[c0:c1].t(), alpha=-1.0)
                first = False
            if side is not None and M > chunk:
                pending = torch.cuda.Event()
                pending.record(side)
                keep.append(lp)
    if pending is not None:
        main.wait_event(pending)
    invert_by_doubling(top)
    del keep
    U = torch.flip(X[:K, :K], (0, 1)).contiguous()
    if not check:
        return U, info
    bad = int(info.item())
    if bad != 0:
        raise torch.linalg.LinAlgError(f"inverse_cholesky_upper: the matrix is not positive definite (pivot <= 0 in diagonal block {bad})")
    return U


def python_column_loop(lookahead, fuse_find_params=True):
    """A drop-in for GPTQ.column_loop (monkeypatch it onto the class) that issues the launches of inc_gptq_quantize_layer one
    by one from Python: [find_params] -> inc_gptq_quant_block[_params] -> inc_gptq_lazy_update (one stream), or with `lookahead`
    the next 128 columns' update on this stream and the rest on a second one (events between them)."""

    def loop(self, w32, Hinv, scale, zero, loop_scale, loop_zero, codes, Q, gs, kernel_gs, blocksize, bits, sym, dynamic_groups, mse):
        N, K = w32.shape
        err = torch.empty((N, QBLOCK), dtype=torch.float32, device=w32.device)
        look = lookahead and K % QBLOCK == 0 and blocksize % QBLOCK == 0 and K >= 3 * QBLOCK and w32.is_cuda
        if look:
            main = torch.cuda.current_stream(w32.device)
            side = torch.cuda.Stream(device=w32.device)
            errs = (err, torch.empty_like(err))
            side.wait_stream(main)  # w32 / Hinv / scales were produced on the main stream
            rest_done = None
            blk = 0
        # find_params fused into the quantisation launch: only when the reference block IS the 128-column block and every group lies
        # inside it -- with a larger reference block the second half's parameters must come from W BEFORE the first half's lazy
        # update (gptq.py:1266-1272 reads the global W)
        fuse_params = (fuse_find_params and dynamic_groups and not mse and blocksize == QBLOCK and gs in (32, 64, QBLOCK)
                       and K % QBLOCK == 0 and loop_scale is scale)
        i1 = 0
        while i1 < K:
            ref_end = min((i1 // blocksize + 1) * blocksize, K)  # end of the reference's block (gptq.py:1250)
            count = min(QBLOCK, ref_end - i1)
            if dynamic_groups and i1 % blocksize == 0 and not fuse_params:
                g_first = -(-i1 // gs)
                g_last = (ref_end - 1) // gs
                if g_last >= g_first:
                    if look and rest_done is not None and (g_last + 1) * gs > i1 + QBLOCK:
                        main.wait_event(rest_done)  # those columns are still being updated by the previous block's remainder
                    ops.gptq_find_params(w32, g_first * gs, gs, g_last - g_first + 1, bits, sym, scale, zero, g_first, mse=mse)

            def quant_block(e):
                if fuse_params:
                    if not ops.gptq_quant_block_params(w32, Hinv, scale, zero, codes, Q, e, i1, count, gs, bits, sym):
                        raise RuntimeError("inc_gptq_quant_block_params refused a full 128-column block")
                else:
                    ops.gptq_quant_block(w32, Hinv, loop_scale, loop_zero, codes, Q, e, i1, count, kernel_gs, bits)

            if not look:
                quant_block(err)
                ops.gptq_lazy_update(w32, Hinv, err, i1, count)
                i1 += count
                continue
            e = errs[blk & 1]
            quant_block(e)
            i2 = i1 + count
            if i2 < K:
                if rest_done is not None:
                    main.wait_event(rest_done)  # rest(b-1) wrote the columns next(b) is about to update (and read Err of b-1)
                nxt_end = min(i2 + QBLOCK, K)
                if not ops.gptq_lazy_update_cols(w32, Hinv, e, i1, count, i2, nxt_end):
                    raise RuntimeError("inc_gptq_lazy_update_cols refused a full 128-column block")
                if nxt_end < K:
                    ready = torch.cuda.Event()
                    ready.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        ops.gptq_lazy_update_cols(w32, Hinv, e, i1, count, nxt_end, K)
                        rest_done = torch.cuda.Event()
                        rest_done.record(side)
            i1 += count
            blk += 1
        if look:
            main.wait_stream(side)  # w32 and both Err buffers are free again

    return loop
