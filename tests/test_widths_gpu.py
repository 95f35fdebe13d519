"""Any feature width on the fast path: the reference builds F + num_indicators input columns
(main.py:111-113) — 131 on ogbn-arxiv, 605 on Reddit, 1436 on Cora — none of them a multiple of 4.

* the fused gather-SpMM on rows padded to whole float4 chunks (indicator columns laid over the chunk that straddles the end of
  X) against gather_rows + the generic aggregation and against the oracle's gcn_conv;
* the gathered-operand GEMMs of the transform-first first layers (ops.linear_fwd_gathered / linear_bwd_weight_gathered)
  against fp64 on the materialised operand;
* the bf16x3 forward GEMM for K up to 192 (arxiv: K = 132) against fp64.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grapes_oracle as O


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).cuda()


def _frontier(rng, n, n_prev):
    hub = np.stack([rng.permutation(n)[:300], np.full(300, 7, np.int64)])
    rnd = rng.integers(0, n, (2, 40000))
    indptr, indices = O.build_csr(np.concatenate([hub, hub[::-1], rnd, rnd[::-1]], axis=1), n)
    prev = rng.permutation(n)[:n_prev].astype(np.int64); prev[0] = 7
    tm = O.TensorMap(n)
    _, batch_nodes, _, local = O.hop_index_pipeline(prev, indptr, indices, tm, n)
    rev = (rng.random(local.shape[1]) < 0.3) | (local[0] == int(np.searchsorted(batch_nodes, 7)))
    ls = np.concatenate([local[0], local[1][rev]]); ld = np.concatenate([local[1], local[0][rev]])
    order = np.lexsort((ld, ls))
    return batch_nodes, ls[order], ld[order]


@pytest.mark.parametrize("F,num_ind", [(128, 3), (602, 3), (1433, 3), (101, 0), (6, 5), (100, 4), (3, 8)])
def test_gather_spmm_any_width(F, num_ind):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(F * 10 + num_ind)
    n = 9000
    batch_nodes, ls, ld = _frontier(rng, n, 200)
    nloc = len(batch_nodes)
    X = _t(rng.standard_normal((n, F)).astype(np.float32))
    Xp, Fl = ops.pad_features(X)
    assert Fl == F and Xp.shape[1] % 4 == 0 and torch.equal(Xp[:, :F], X) and float(Xp[:, F:].abs().sum()) == 0.0
    ids = _t(batch_nodes, torch.int32)
    epoch = (1 << 23) + 5                                           # a host-side epoch (sets the sign bit of the int32 code)
    code_np = (((epoch << 8) | rng.integers(0, 1 << max(num_ind, 1), n)) & 0xffffffff).astype(np.uint32).view(np.int32)
    code_np[::5] = (((epoch - 1) << 8) | 0xff) & 0xffffffff if False else code_np[::5]
    code = _t(code_np)
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    plain = ops.PreparedGraph(_t(ls, torch.int32), _t(ld, torch.int32), nloc, status=st, src_grouped=True)
    heads = ops.PreparedGraph(_t(ls, torch.int32), _t(ld, torch.int32), nloc, status=st, src_grouped=True, head_ids=ids)
    kp = (F + num_ind + 3) // 4 * 4
    a = ops.gcn_aggregate_gather(Xp, ids, plain, code if num_ind else None, epoch, num_ind, F=F)
    b = ops.gcn_aggregate_gather(Xp, ids, heads, code if num_ind else None, epoch, num_ind, F=F)
    assert a.shape == (nloc, kp) and torch.equal(a, b)
    assert float(a[:, F + num_ind:].abs().sum()) == 0.0                       # padding columns are zeros
    xg = ops.gather_rows(X, ids, code if num_ind else None, epoch, num_ind)   # [n, F + num_ind] exact width
    ref = O.gcn_conv(xg.cpu(), torch.eye(F + num_ind), None, torch.from_numpy(np.stack([ls, ld])))
    assert float((b[:, :F + num_ind].cpu() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("n,F,num_ind,fo", [(5000, 602, 3, 256), (700, 1433, 3, 256), (33000, 602, 0, 256), (130, 37, 2, 64),
                                            (1025, 1433, 0, 256)])
def test_gathered_operand_gemms(n, F, num_ind, fo):
    """H = feat(ids) Wᵀ and dW = dHᵀ feat(ids) with the operand read through the id list, against fp64 on the materialised
    operand; n on the device (capacity-padded), few-row split-K and many-row forms."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(n + F)
    N = 50000
    X = _t(rng.standard_normal((N, F)).astype(np.float32))
    Xp, _ = ops.pad_features(X)
    cap = n + 37
    ids = _t(rng.integers(0, N, cap), torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    epoch = 77
    code = _t(((epoch << 8) | rng.integers(0, 1 << max(num_ind, 1), N)).astype(np.int32))
    code[::4] = ((epoch - 1) << 8) | 0xff                                   # stale epoch: indicators read as zero
    K, kp = F + num_ind, (F + num_ind + 3) // 4 * 4
    W = _t((rng.standard_normal((fo, K)) / np.sqrt(K)).astype(np.float32))
    Wp = torch.zeros(fo, kp, device="cuda"); Wp[:, :K] = W
    if kp > K:
        Wp[:, K:] = 123.0                                                   # padding columns of W must be ignored (x 0)
    h = ops.linear_fwd_gathered(Xp, F, ids, Wp, code if num_ind else None, epoch, num_ind, d_n=d_n)
    feat = ops.gather_rows(X, ids[:n].contiguous(), code if num_ind else None, epoch, num_ind).cpu().double()
    ref = feat @ W.cpu().double().t()
    scale = float((feat.abs() @ W.cpu().double().abs().t()).max())
    assert float((h[:n].cpu().double() - ref).abs().max()) <= 2e-6 * scale
    dh = _t(rng.standard_normal((cap, fo)).astype(np.float32))
    dW = torch.full((fo, kp), 7.0, device="cuda")
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code if num_ind else None, epoch, num_ind, d_n=d_n, accumulate=False)
    refw = dh[:n].cpu().double().t() @ feat
    scw = float((dh[:n].cpu().double().abs().t() @ feat.abs()).max())
    assert float((dW[:, :K].cpu().double() - refw).abs().max()) <= 2e-6 * scw
    assert float(dW[:, K:].abs().sum()) == 0.0
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code if num_ind else None, epoch, num_ind, d_n=d_n, accumulate=True)
    assert float((dW[:, :K].cpu().double() - 2 * refw).abs().max()) <= 4e-6 * scw


@pytest.mark.parametrize("n,F,num_ind,fo", [(5000, 602, 3, 256), (700, 1433, 3, 256), (77015, 602, 3, 256), (130, 37, 2, 64),
                                            (1025, 1433, 0, 256), (129, 602, 0, 132)])
def test_gathered_operand_gemms_on_the_bf16_pipe(n, F, num_ind, fo):
    """The bf16x3 tiled forms (csrc/gemm_tiled_split.hip) of H = feat(ids) Wᵀ and dW = dHᵀ feat(ids): against fp64 on the
    materialised operand at the accuracy of the fp32-MFMA kernels (they must be no worse), W taken from an UNPADDED parameter
    through its split image, capacity-padded rows, masked indicator bits in the backward."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    assert ops.split_gathered_available(fo)
    rng = np.random.default_rng(n + F + 1)
    N = 60000
    X = _t(rng.standard_normal((N, F)).astype(np.float32))
    Xp, _ = ops.pad_features(X)
    cap = n + 41
    ids = _t(rng.integers(0, N, cap), torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    epoch = 31
    code = _t(((epoch << 8) | rng.integers(0, 1 << max(num_ind, 1), N)).astype(np.int32))
    K, kp = F + num_ind, (F + num_ind + 3) // 4 * 4
    W = _t((rng.standard_normal((fo, K)) / np.sqrt(K)).astype(np.float32))
    img = ops.weight_split_image(W)
    Wp = torch.zeros(fo, kp, device="cuda"); Wp[:, :K] = W
    h = ops.linear_fwd_gathered(Xp, F, ids, Wp, code if num_ind else None, epoch, num_ind, d_n=d_n, w_image=img)
    h32 = ops.linear_fwd_gathered(Xp, F, ids, Wp, code if num_ind else None, epoch, num_ind, d_n=d_n)
    feat = ops.gather_rows(X, ids[:n].contiguous(), code if num_ind else None, epoch, num_ind).cpu().double()
    ref = feat @ W.cpu().double().t()
    scale = float((feat.abs() @ W.cpu().double().abs().t()).max())
    e_split = float((h[:n].cpu().double() - ref).abs().max()) / scale
    e_fp32 = float((h32[:n].cpu().double() - ref).abs().max()) / scale
    assert e_split <= 5e-7 and e_fp32 <= 5e-7, (e_split, e_fp32)     # relative to sum |a||b| (fp32 eps = 6e-8; K up to 1436)
    if n >= 8192:
        # many rows: the split-K entry degenerates to one slab, i.e. to the plain kernel — bit for bit (and needs no sum launch)
        h1 = torch.full_like(h, 7.0)
        ws = torch.empty(int(ops.lib().grapes_linear_fwd_gathered_split_k_workspace_bytes(cap, kp, fo)) + 16, dtype=torch.uint8, device="cuda")
        rc = ops.lib().grapes_linear_fwd_gathered_split_k(Xp.data_ptr(), F, Xp.shape[1], ids.data_ptr(), code.data_ptr() if num_ind else None,
                                                          epoch, None, num_ind, img.data_ptr(), h1.data_ptr(), cap, d_n.data_ptr(), fo,
                                                          ws.data_ptr() + (-ws.data_ptr()) % 16, torch.cuda.current_stream().cuda_stream)
        hp = torch.full_like(h, 5.0)          # (the plain entry: ops.linear_fwd_gathered itself runs the split-tail form)
        assert ops.lib().grapes_linear_fwd_gathered_split(Xp.data_ptr(), F, Xp.shape[1], ids.data_ptr(), code.data_ptr() if num_ind else None,
                                                          epoch, None, num_ind, img.data_ptr(), hp.data_ptr(), cap, d_n.data_ptr(), fo,
                                                          torch.cuda.current_stream().cuda_stream) == 0
        assert rc == 0 and torch.equal(h1[:n], hp[:n])
    # backward with an indicator mask (bits 0 and num_ind - 1 only)
    mask = (1 | (1 << (num_ind - 1))) if num_ind else 0
    featm = feat.clone()
    if num_ind:
        keepcols = [j for j in range(num_ind) if (mask >> j) & 1]
        for j in range(num_ind):
            if j not in keepcols:
                featm[:, F + j] = 0.0
    dh = _t(rng.standard_normal((cap, fo)).astype(np.float32))
    dW = torch.full((fo, kp), 3.0, device="cuda")
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code if num_ind else None, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True)
    refw = dh[:n].cpu().double().t() @ featm
    scw = float((dh[:n].cpu().double().abs().t() @ featm.abs()).max())
    assert float((dW[:, :K].cpu().double() - refw).abs().max()) <= 1e-6 * scw
    assert float(dW[:, K:].abs().sum()) == 0.0
    ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, code if num_ind else None, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True,
                                   accumulate=True)
    assert float((dW[:, :K].cpu().double() - 2 * refw).abs().max()) <= 2e-6 * scw
    if kp != K:
        # the parameter's own [f_out, K] layout written by the slab sum (no padded buffer + strided copy): the same numbers, and
        # accumulation on top of them; the padded fp32 copy of W out of the image launch equals the zero-padded copy
        dWp = torch.full((fo, kp), 3.0, device="cuda")
        ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dWp, code if num_ind else None, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True)
        dWu = torch.full((fo, K), 5.0, device="cuda")
        ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dWu, code if num_ind else None, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True)
        assert torch.equal(dWu, dWp[:, :K])
        ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dWu, code if num_ind else None, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True,
                                       accumulate=True)
        assert torch.equal(dWu, dW[:, :K])
        wpad = torch.full((fo, kp), 9.0, device="cuda")
        img2 = ops.weight_split_image(W, w_pad=wpad)
        assert torch.equal(img2, img) and torch.equal(wpad, Wp)


@pytest.mark.parametrize("n", [9000, 12800, 23300, 33000, 76500, 77015])
def test_forward_gemm_with_a_split_tail_for_one_and_two_nets(n):
    """grapes_linear_fwd_gathered_split_tail: tiles of whole rounds (256 resident workgroups) are BIT-IDENTICAL to
    grapes_linear_fwd_gathered_split; the tiles of the last partial round are cut along K into pieces and summed in piece order —
    equal to fp64 at the accuracy asserted for the plain kernel.  One net, and two nets over the same rows (with / without indicator
    columns, their own weights); row counts that give no tail cut (S = 1), a tail of 2 .. 8 pieces, a partial last tile, a capacity
    above the live count."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    F, ni, fo, N = 602, 3, 256, 90000
    rng = np.random.default_rng(n)
    X = _t(rng.standard_normal((N, F)).astype(np.float32))
    Xp, _ = ops.pad_features(X)
    cap = n + 300
    ids = _t(rng.integers(0, N, cap), torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    epoch = 5
    code = _t(((epoch << 8) | rng.integers(0, 1 << ni, N)).astype(np.int32))
    Wa = _t((rng.standard_normal((fo, F + ni)) / np.sqrt(F)).astype(np.float32)); Wb = _t((rng.standard_normal((fo, F)) / np.sqrt(F)).astype(np.float32))
    ia, ib = ops.weight_split_image(Wa), ops.weight_split_image(Wb)
    pa = torch.zeros(fo, 608, device="cuda"); pa[:, :F + ni] = Wa
    pb = torch.zeros(fo, 604, device="cuda"); pb[:, :F] = Wb
    plain_a = torch.empty((cap, fo), device="cuda"); plain_b = torch.empty((cap, fo), device="cuda")
    L = ops.lib()
    st = torch.cuda.current_stream().cuda_stream
    assert L.grapes_linear_fwd_gathered_split(Xp.data_ptr(), F, Xp.shape[1], ids.data_ptr(), code.data_ptr(), epoch, None, ni, ia.data_ptr(),
                                              plain_a.data_ptr(), cap, d_n.data_ptr(), fo, st) == 0
    assert L.grapes_linear_fwd_gathered_split(Xp.data_ptr(), F, Xp.shape[1], ids.data_ptr(), None, epoch, None, 0, ib.data_ptr(),
                                              plain_b.data_ptr(), cap, d_n.data_ptr(), fo, st) == 0
    feat = ops.gather_rows(X, ids[:n].contiguous(), code, epoch, ni).cpu().double()
    for nets in (1, 2):
        outs = ops.linear_fwd_gathered_tail(Xp, F, ids, [ia, ib][:nets], fo, [code, None][:nets], [ni, 0][:nets], epoch=epoch, d_n=d_n)
        torch.cuda.synchronize()
        ntiles = (n + 127) // 128
        units = ntiles * nets
        t_full = units // 256 * 256
        for q, (o, pl, W, k) in enumerate(zip(outs, (plain_a, plain_b), (Wa, Wb), (F + ni, F))):
            # the tiles of whole rounds: units u = tile * nets + q < t_full
            whole = [t for t in range(ntiles) if t * nets + q < t_full]
            if whole:
                rows = torch.cat([torch.arange(t * 128, min((t + 1) * 128, n)) for t in whole]).cuda()
                assert torch.equal(o[rows], pl[rows])
            ref = feat[:, :k] @ W.cpu().double().t()
            scale = float((feat[:, :k].abs() @ W.cpu().double().abs().t()).max())
            assert float((o[:n].cpu().double() - ref).abs().max()) <= 5e-7 * scale
            assert float((o[:n] - pl[:n]).abs().max()) <= 2e-6 * scale
    # the wrapper's one-net path (what the step calls)
    h = ops.linear_fwd_gathered(Xp, F, ids, pa, code, epoch, ni, d_n=d_n, w_image=ia)
    assert torch.equal(h[:n], ops.linear_fwd_gathered_tail(Xp, F, ids, [ia], fo, [code], [ni], epoch=epoch, d_n=d_n)[0][:n])


def test_weight_images_of_three_layers_in_one_launch_equal_three_launches():
    """grapes_weight_split_images (the step's first layers refreshed together) writes the images and padded copies of
    grapes_weight_split_image[_padded], bit for bit; weights of different shapes, a view with a row stride, one without a copy."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(5)
    big = _t(rng.standard_normal((256, 700)).astype(np.float32))
    ws = [big[:, :605], _t(rng.standard_normal((256, 602)).astype(np.float32)), _t(rng.standard_normal((96, 33)).astype(np.float32))]
    pads = [torch.full((256, 608), 7.0, device="cuda"), None, torch.full((96, 36), 7.0, device="cuda")]
    nb = lambda w: int(ops.lib().grapes_weight_split_image_bytes(int(w.shape[1])))
    imgs = [torch.full((nb(w),), 0xAB, dtype=torch.uint8, device="cuda") for w in ws]
    ops.weight_split_images(ws, imgs, pads)
    for w, im, wp in zip(ws, imgs, pads):
        ref_pad = None if wp is None else torch.full_like(wp, 7.0)
        ref = ops.weight_split_image(w, w_pad=ref_pad)
        assert torch.equal(im, ref)
        if wp is not None:
            assert torch.equal(wp, ref_pad)
            assert torch.equal(wp[:, :w.shape[1]], w) and float(wp[:, w.shape[1]:].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        ops.weight_split_images(ws, imgs[:2])


@pytest.mark.parametrize("n,K,N", [(40000, 132, 256), (5000, 160, 256), (2100, 192, 96), (37501, 104, 256)])
def test_split_gemm_wide_k(n, K, N):
    """The bf16x3 forward GEMM with the fused head projection for K up to 192 (arxiv's 128 + 3 -> 132) against fp64."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(K)
    x = _t(rng.standard_normal((n, K)).astype(np.float32))
    w = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = _t(rng.standard_normal(N).astype(np.float32))
    hw = _t(rng.standard_normal(N).astype(np.float32))
    assert ops.split_gemm_available(n, K, N) or K > 112      # (the dW side of the pair stops at 112; the forward kernel not)
    out, head = ops.linear_bias_act_head_fwd(x, w, b, True, hw)
    ref = torch.relu(x.cpu().double() @ w.cpu().double().t() + b.cpu().double())
    scale = float((x.cpu().double().abs() @ w.cpu().double().abs().t()).max())
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-6 * scale
    refh = ref @ hw.cpu().double()
    assert float((head.view(-1).cpu().double() - refh).abs().max()) <= 1e-5 * max(1.0, float(refh.abs().max()))


@pytest.mark.parametrize("n_rows,n_cand,k,philox", [(30000, 21000, 256, True), (30000, 21000, 256, False), (900, 300, 512, True),
                                                    (70000, 69999, 1, True), (300, 7, 3, False)])
def test_draw_fused_with_the_logit_aggregation(n_rows, n_cand, k, philox):
    """ops.gumbel_topk(agg=...) — the sampler net's 1-wide aggregation and the key computation in one launch — against
    gcn_aggregate_fwd + gumbel_topk: logits, masks, kept positions / ids, log-probabilities bit-identical; statistics to fp32
    rounding; device-side counts (capacity-padded buffers), keep-all (k >= n) and Philox / given uniforms."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(n_rows + n_cand + k)
    cap = n_rows + 100
    e = 3 * n_rows
    src = np.sort(rng.integers(0, max(n_rows // 20, 1), e)); dst = rng.integers(0, n_rows, e)      # source-grouped, hubby
    key = np.unique(src.astype(np.int64) * n_rows + dst); src, dst = key // n_rows, key % n_rows
    keep = src != dst; src, dst = src[keep], dst[keep]
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_rows = torch.tensor([n_rows], dtype=torch.int32, device="cuda")
    prep = ops.PreparedGraph(_t(src, torch.int32), _t(dst, torch.int32), cap, d_n=d_rows, status=st, src_grouped=True, items_fwd=False)
    hw = _t(rng.standard_normal(cap).astype(np.float32) * 2)
    bias = _t(np.array([0.3], np.float32))
    cand_rows = np.sort(rng.permutation(n_rows)[:n_cand])
    nbl = torch.zeros(cap, dtype=torch.int32, device="cuda"); nbl[:n_cand] = _t(cand_rows, torch.int32)
    cp = np.full(cap, -1, np.int32); cp[cand_rows] = np.arange(n_cand)
    cand_pos = _t(cp)
    ids = _t(rng.permutation(10 * cap)[:cap].astype(np.int32))
    d_nc = torch.tensor([n_cand], dtype=torch.int32, device="cuda")
    uni = None if philox else _t(rng.random(cap, dtype=np.float32))
    prefix = _t(np.arange(5, dtype=np.int32))
    kw = dict(logit_index=nbl, candidate_ids=ids, n=cap, d_n=d_nc, uniforms=uni, philox_seed=11, prefix_ids=prefix, want_keys=True)
    off_a = torch.tensor([9], dtype=torch.int64, device="cuda"); off_b = off_a.clone()
    logits = ops.gcn_aggregate_fwd(hw.view(-1, 1), prep, bias, False)
    a = ops.gumbel_topk(logits.view(-1), k, d_philox_offset=off_a if philox else None, **kw)
    b = ops.gumbel_topk(None, k, d_philox_offset=off_b if philox else None, agg=(hw, prep, bias, cand_pos), **kw)
    torch.cuda.synchronize()
    assert int(st) == 0
    assert torch.equal(b["logits"][:n_rows], logits[:n_rows])
    kk = min(k, n_cand)
    assert int(a["kept_count"]) == int(b["kept_count"]) == kk and int(a["union_count"]) == int(b["union_count"]) == 5 + kk
    assert torch.equal(a["mask"][:n_cand], b["mask"][:n_cand])
    assert torch.equal(a["kept_pos"][:kk], b["kept_pos"][:kk]) and torch.equal(a["kept_ids"][:kk], b["kept_ids"][:kk])
    assert torch.equal(a["union_ids"][:5 + kk], b["union_ids"][:5 + kk])
    assert torch.equal(a["log_prob"][:n_cand], b["log_prob"][:n_cand])
    if k < n_cand:
        assert torch.equal(a["keys"][:n_cand], b["keys"][:n_cand])
    assert torch.allclose(a["stats"], b["stats"], rtol=1e-6, atol=1e-7)
    assert int(off_a) == int(off_b)


@pytest.mark.parametrize("n,F,num_ind", [(9000, 602, 3), (9000, 1433, 3), (700, 1433, 3)])
def test_gathered_operand_gemms_hold_fp32_accuracy_over_a_wide_dynamic_range(n, F, num_ind):
    """`gemm_tsplit_fwd_k` / `gemm_tsplit_dw_k` (the bf16x3 tiled GEMMs of the transform-first layers, K = 608 /
    1436: the longest accumulation chains in the library; n = 700 takes the split-K form) on operands that are NOT N(0,1):
    per-column (forward) / per-row (dW) magnitudes spanning 1e-20 ... 1e20 with the other operand scaled the other way so that
    the products stay finite, exact zeros (rows, columns, single entries), one huge entry (1e30) and tiny ones (1e-30).  Each
    output's error against fp64 is measured relative to ITS OWN sum |a||b| — a mixed-magnitude sum is where a dropped split
    term would show — and must be no larger than the fp32-MFMA kernel's on the same data (+5 %), and below 1e-6."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    fo = 256
    assert ops.split_gathered_available(fo)
    rng = np.random.default_rng(F + n)
    N = 20000
    K, kp = F + num_ind, (F + num_ind + 3) // 4 * 4
    a_col = rng.uniform(-20, 20, F)                      # forward: column k of X at 10^a, column k of W at 10^(-a + b)
    X = rng.standard_normal((N, F)) * 10.0 ** a_col
    X[rng.integers(0, N, 50)] = 0.0                      # zero rows
    X[:, rng.integers(0, F, 5)] = 0.0                    # zero columns
    X[rng.integers(0, N, 2000), rng.integers(0, F, 2000)] = 0.0
    W = rng.standard_normal((fo, K)) / np.sqrt(K)
    W[:, :F] *= 10.0 ** (-a_col + rng.uniform(-3, 3, F))
    W[3, :] = 0.0
    W[5, 7] = -1e-30
    cap = n + 41
    ids_np = rng.integers(0, N, cap)
    X[ids_np[n // 3], 11] = 1e30 * 10.0 ** min(0.0, a_col[11] - 20)   # one huge entry (its products stay below fp32 max)
    X = _t(X.astype(np.float32)); W = _t(W.astype(np.float32))
    assert bool(torch.isfinite(X).all()) and bool(torch.isfinite(W).all())
    Xp, _ = ops.pad_features(X)
    ids = _t(ids_np, torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    epoch = 5
    code = _t(((epoch << 8) | rng.integers(0, 1 << num_ind, N)).astype(np.int32))
    img = ops.weight_split_image(W)
    Wp = torch.zeros(fo, kp, device="cuda"); Wp[:, :K] = W
    feat = ops.gather_rows(X, ids[:n].contiguous(), code, epoch, num_ind).cpu().double()
    ref = feat @ W.cpu().double().t()
    mag = (feat.abs() @ W.cpu().double().abs().t()).clamp_min(1e-300)
    assert bool(torch.isfinite(ref.float()).all())
    h = ops.linear_fwd_gathered(Xp, F, ids, Wp, code, epoch, num_ind, d_n=d_n, w_image=img)          # bf16x3 (split-K when n < 8192)
    h32 = ops.linear_fwd_gathered(Xp, F, ids, Wp, code, epoch, num_ind, d_n=d_n)                      # fp32 MFMA
    assert bool(torch.isfinite(h[:n]).all())
    e = {k: (v[:n].cpu().double() - ref).abs() / mag for k, v in (("split", h), ("fp32", h32))}
    emax = {k: float(v.max()) for k, v in e.items()}; erms = {k: float((v ** 2).mean().sqrt()) for k, v in e.items()}
    assert emax["split"] <= 1.05 * emax["fp32"] + 1e-9 and erms["split"] <= 1.05 * erms["fp32"] + 1e-10, (emax, erms)
    # (absolute: a K = 1436 fp32 accumulation of mixed magnitudes — measured worst output 1.05e-6 for the split kernel, 1.7e-6 for the
    # fp32-MFMA kernel, of the output's own sum |a||b|)
    print(f"[wide range fwd] n={n} K={K}: max split {emax['split']:.3e} fp32 {emax['fp32']:.3e}; rms split {erms['split']:.3e} fp32 {erms['fp32']:.3e}")
    assert emax["split"] < 3e-6, emax
    # dW = dH^T feat(ids): the contraction runs over the ROWS — row r of dH at 10^(-a_r + b), feature row ids[r] at 10^a_r
    a_node = rng.uniform(-20, 20, N)
    X2 = rng.standard_normal((N, F)) * 10.0 ** a_node[:, None]
    X2[rng.integers(0, N, 50)] = 0.0
    X2[rng.integers(0, N, 2000), rng.integers(0, F, 2000)] = 0.0
    dh = rng.standard_normal((cap, fo)) * 10.0 ** (-a_node[ids_np] + rng.uniform(-3, 3, cap))[:, None]
    dh[rng.integers(0, n, 20)] = 0.0
    dh[:, 9] = 0.0
    # (the 0/1 indicator columns meet dH at 10^(-a_r): keep their sums finite — nodes with a large negative exponent carry no bits)
    code2 = ((epoch << 8) | np.where(a_node > -10, rng.integers(0, 1 << num_ind, N), 0)).astype(np.int32)
    X2 = _t(X2.astype(np.float32)); dh = _t(dh.astype(np.float32)); code2 = _t(code2)
    assert bool(torch.isfinite(X2).all()) and bool(torch.isfinite(dh).all())
    X2p, _ = ops.pad_features(X2)
    feat2 = ops.gather_rows(X2, ids[:n].contiguous(), code2, epoch, num_ind).cpu().double()
    refw = dh[:n].cpu().double().t() @ feat2
    magw = (dh[:n].cpu().double().abs().t() @ feat2.abs()).clamp_min(1e-300)
    assert bool(torch.isfinite(refw.float()).all())
    full = (1 << num_ind) - 1
    got = {}
    for name, split in (("split", True), ("fp32", False)):
        dW = torch.full((fo, kp), 3.0, device="cuda")
        ops.linear_bwd_weight_gathered(dh, X2p, F, ids, dW, code2, epoch, num_ind, d_n=d_n, ind_mask=full, split=split)
        assert float(dW[:, K:].abs().sum()) == 0.0 and bool(torch.isfinite(dW).all())
        got[name] = (dW[:, :K].cpu().double() - refw).abs() / magw
    emax = {k: float(v.max()) for k, v in got.items()}; erms = {k: float((v ** 2).mean().sqrt()) for k, v in got.items()}
    # Measured (MI355X, round 4): rms 1.16e-8 against the fp32-MFMA kernel's 1.13e-8, WORST output 2.6e-7 against 1.1e-7 of its
    # own sum |a||b| — the weight gradient adds its slab partials (up to 768 of them) in fp32, and with mixed magnitudes one
    # output in 3.7e5 lands two ulps of its magnitude sum further out than the fp32 kernel's worst.  The bound asserted is what
    # "fp32 accuracy" means here: an rms below half an ulp of the magnitude sum (4e-8; within 1.6 x the fp32 kernel's) and every
    # output within 4 ulps (4 x 2^-23) of its magnitude sum.
    print(f"[wide range dW] n={n} K={K}: max split {emax['split']:.3e} fp32 {emax['fp32']:.3e}; rms split {erms['split']:.3e} fp32 {erms['fp32']:.3e}")
    # (n = 700, K = 1436 — the few-row case, 64 slabs of a handful of K steps each: rms 2.7e-8 against 1.8e-8.)
    assert erms["split"] <= 1.6 * erms["fp32"] + 1e-10 and erms["split"] <= 4e-8, (emax, erms)
    assert emax["split"] <= 4 * 2.0 ** -23, emax


@pytest.mark.parametrize("f,live_frac", [(256, 1.0), (256, 0.6), (64, 1.0), (100, 0.9), (20, 1.0)])
def test_record_driven_aggregation_is_bit_identical_to_the_csr_walk(f, live_frac):
    """grapes_gcn_aggregate_fwd_rec (head records over LOCAL ids: one dependent trip per row, pairs of rows per resident
    wavefront) against grapes_gcn_aggregate_fwd / _fwd_head (rowptr -> csr -> rows): the same products in the same order, so
    the activations, the head products and the gate bits are equal BIT FOR BIT — on a frontier-shaped graph with rows of 0..4
    entries, 5..16 entries and a hub of hundreds, with a device-side row count below the capacity, bias + ReLU."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from grapes_amd import ops
    rng = np.random.default_rng(f)
    batch_nodes, ls, ld = _frontier(rng, 30000, 400)
    n = len(batch_nodes)
    cap = n + 57
    d_n = torch.tensor([max(1, int(n * live_frac))], dtype=torch.int32, device="cuda")
    nl = int(d_n.item())
    keep = (ls < nl) & (ld < nl)
    src, dst = _t(ls[keep], torch.int32), _t(ld[keep], torch.int32)
    iota = torch.arange(cap, dtype=torch.int32, device="cuda")
    plain = ops.PreparedGraph(src, dst, cap, d_n=d_n, src_grouped=True, items_fwd=False)
    recs = ops.PreparedGraph(src, dst, cap, d_n=d_n, src_grouped=True, items_fwd=False, head_ids=iota, head_local=True)
    assert recs.row_head is not None and recs.head_local and not plain.head_local
    lens = (recs.rowptr_t[1:nl + 1] - recs.rowptr_t[:nl]).cpu().numpy()
    assert (lens <= 4).any() and (lens > 16).any()                          # short rows, and at least the hub
    h = _t(rng.standard_normal((cap, f)).astype(np.float32))
    bias = _t(rng.standard_normal(f).astype(np.float32))
    w2 = _t(rng.standard_normal(f).astype(np.float32))
    for relu in (True, False):
        a = ops.gcn_aggregate_fwd(h, plain, bias, relu)
        b = ops.gcn_aggregate_fwd(h, recs, bias, relu)
        assert torch.equal(a[:nl], b[:nl]), (f, relu)
    ra = ops.gcn_aggregate_fwd_head(h, plain, bias, True, w2, want_bits=True)
    rb = ops.gcn_aggregate_fwd_head(h, recs, bias, True, w2, want_bits=True)
    for x, y in zip(ra, rb):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x[:nl], y[:nl]), f
    # ... and against the oracle's GCNConv (identity weight: the aggregation alone) at 1e-5
    ref = O.gcn_conv(h[:nl].cpu(), torch.eye(f), bias.cpu(), torch.stack([src.cpu().long(), dst.cpu().long()]))
    got = ops.gcn_aggregate_fwd(h, recs, bias, False)[:nl].cpu()
    assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def _planes_case(n, F, num_ind):
    from grapes_amd import ops
    fo = 256
    rng = np.random.default_rng(n + F)
    N = 40000
    X = _t((rng.standard_normal((N, F)) * 10.0 ** rng.uniform(-3, 3, (N, 1))).astype(np.float32))
    Xp, _ = ops.pad_features(X)
    cap = n + 29
    ids = _t(rng.integers(0, N, cap), torch.int32)
    d_n = torch.tensor([n], dtype=torch.int32, device="cuda")
    epoch = 9
    code = _t(((epoch << 8) | rng.integers(0, 1 << max(num_ind, 1), N)).astype(np.int32))
    K, kp = F + num_ind, (F + num_ind + 3) // 4 * 4
    W = _t((rng.standard_normal((fo, K)) / np.sqrt(K)).astype(np.float32))
    img = ops.weight_split_image(W)
    Wp = torch.zeros(fo, kp, device="cuda"); Wp[:, :K] = W
    dh = _t(rng.standard_normal((cap, fo)).astype(np.float32))
    mask = (1 | (1 << (num_ind - 1))) if num_ind else 0
    cd = code if num_ind else None

    def run():
        h = ops.linear_fwd_gathered(Xp, F, ids, Wp, cd, epoch, num_ind, d_n=d_n, w_image=img)
        dW = torch.full((fo, kp), 2.0, device="cuda")
        ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, cd, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True)
        ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dW, cd, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True, accumulate=True)
        dWu = None
        if kp != K:
            dWu = torch.full((fo, K), 5.0, device="cuda")
            ops.linear_bwd_weight_gathered(dh, Xp, F, ids, dWu, cd, epoch, num_ind, d_n=d_n, ind_mask=mask, split=True)
        torch.cuda.synchronize()
        return h[:n].clone(), dW.clone(), dWu

    a = run()
    planes = ops.FeaturePlanes(Xp)
    b = run()
    planes.close()
    c = run()
    for x, y, z in zip(a, b, c):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y) and torch.equal(x, z)
    assert bool(torch.isfinite(a[0]).all()) and float(a[0].abs().max()) > 0


@pytest.mark.parametrize("n,F,num_ind", [(9000, 602, 3), (700, 1433, 3), (9000, 600, 0), (33000, 37, 2)])
def test_gathered_gemms_from_presplit_planes_are_bit_identical(n, F, num_ind):
    """grapes_feature_split_planes + _register (an A/B form of the DIAGNOSTIC build: it measured slower and does not ship): the
    gathered-operand bf16x3 GEMMs read the three bf16 planes of the rows they gather instead of splitting them in their K loops —
    the same split, done once: forward (plain and split-K) and weight gradient (padded and parameter layout, with an indicator
    mask, accumulating) are equal BIT FOR BIT with and without the planes; unregistering restores the in-loop path.  Runs in a
    child process on libgrapes_hip_diag.so."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r}); "
            f"import test_widths_gpu as T; T._planes_case({n}, {F}, {num_ind}); print('child ok')")
    env = dict(os.environ, GRAPES_DIAG="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
