"""GPU parity of the non-conv kernels against the golden vectors made from the
reference's own classes (tests/golden/make_golden.py) and against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.helpers import golden

pytestmark = pytest.mark.gpu


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ VQ
def _make_quantizer(K, D, ema=True, bdt=True):
    from crank_amd.net.module.flat import FlatModel
    from crank_amd.net.module.vqvae2 import Quantizer

    class Holder(FlatModel):
        def touch_codebook(self):
            pass

    h = Holder()
    q = Quantizer(h, "", D, K, ema_flag=ema, bdt_flag=bdt)
    ents = q.entries(0)
    h._alloc(ents, q.n_params, "cuda")
    h._bufs.update(q.make_buffers("cuda"))
    return h, q


def test_vq_indices_bit_exact_and_ema_vs_reference_quantizer():
    fx = golden("quantizer.npz")
    h, q = _make_quantizer(512, 64)
    q.weight.copy_(cu(fx["init_weight"]))
    q.ema_w.copy_(cu(fx["init_ema_w"]))
    q.ema_size.copy_(cu(fx["init_ema_size"]))
    for it in range(3):
        e, qx, idx = q(cu(fx[f"x{it}"]), use_ema=True)
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), fx[f"idx{it}"]), f"indices differ at iteration {it}"
        np.testing.assert_allclose(e.cpu().numpy(), fx[f"e{it}"], rtol=1e-6, atol=0)
        np.testing.assert_allclose(qx.cpu().numpy(), fx[f"qx{it}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(q.ema_size.cpu().numpy(), fx[f"ema_size{it}"], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(q.ema_w.cpu().numpy(), fx[f"ema_w{it}"], rtol=2e-5, atol=2e-6)
        w, wr = q.weight.cpu().numpy(), fx[f"w{it}"]
        np.testing.assert_allclose(w, wr, rtol=1e-4, atol=1e-5 * np.abs(wr).max())
    e, qx, idx = q(cu(fx["x3"]), use_ema=False)
    assert np.array_equal(idx.cpu().numpy(), fx["idx3"])
    np.testing.assert_allclose(q.weight.cpu().numpy(), fx["w3"], rtol=1e-4, atol=1e-5 * np.abs(fx["w3"]).max())


def test_vq_exact_ties_pick_lowest_index():
    fx = golden("quantizer.npz")
    h, q = _make_quantizer(512, 64, ema=False, bdt=False)
    q.weight.copy_(cu(fx["tie_w"]))
    q.training = False
    e, qx, idx = q(cu(fx["tie_x"]))
    got, ref = idx.cpu().numpy(), fx["tie_idx"]
    # rows built on the duplicated code must resolve to index 7 (never 100 / 300)
    assert (got[0, :9] == 7).all(), got[0, :9]
    assert np.array_equal(got, ref), np.argwhere(got != ref)
    np.testing.assert_array_equal(e.cpu().numpy(), fx["tie_e"])


def test_vq_indices_bit_exact_at_benchmark_size_vs_reference_quantizer():
    """N = 32 000 frames (BASELINE configs[1]) against indices produced by the REFERENCE's Quantizer
    (tests/golden/quantizer_full.npz: codebook after two EMA updates, 64 never-used codes at ~1e5 - quirk Q2 - 447
    codes in use): every one of the 32 000 indices identical, gathered vectors exact."""
    from crank_amd import ops

    fx = golden("quantizer_full.npz")
    B, D, T = [int(v) for v in fx["x_shape_BDT"]]
    x = np.random.RandomState(int(fx["x_seed"])).standard_normal((B, D, T)).astype(np.float32)  # the generator's (B,D,T) layout
    xk = cu(x.transpose(0, 2, 1))
    w = cu(fx["codebook"])
    e, qx, idx = ops.vq_apply(xk, w)
    torch.cuda.synchronize()
    got, ref = idx.cpu().numpy(), fx["idx"].astype(np.int64)
    assert got.shape == ref.shape == (B, T)
    assert np.array_equal(got, ref), f"{(got != ref).sum()} of {got.size} indices differ from the reference Quantizer"
    assert torch.equal(e, w[idx])
    assert len(np.unique(got)) > 400


def test_vq_full_size_properties():
    """BASELINE size (N=32000, K=512, D=64): every chosen code is the fp64 nearest (up to
    rounding), gathering is exact, quantising twice is idempotent, STE passes gradients."""
    from crank_amd import ops

    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(64, 500, 64, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(512, 64, generator=g) * 0.7).cuda()
    e, qx, idx = ops.vq_apply(x, w)
    d = torch.cdist(x.detach().reshape(-1, 64).double(), w.double()) ** 2
    best = d.min(dim=1).values
    chosen = d.gather(1, idx.reshape(-1, 1)).squeeze(1)
    assert ((chosen - best) <= 1e-4 * (1 + best)).all()
    assert (idx.reshape(-1) == d.argmin(1)).float().mean().item() > 0.9999
    assert torch.equal(e, w[idx])
    e2, _, idx2 = ops.vq_apply(e, w)
    assert torch.equal(idx2, idx) and torch.equal(e2, e)
    qx.backward(torch.ones_like(qx))
    assert torch.equal(x.grad, torch.ones_like(x))


def _vq_flags(reset=False):
    import ctypes

    from crank_amd import _lib

    out = (ctypes.c_ulonglong * 3)()
    _lib.check(_lib.lib().crk_debug_vq_flags(out, 1 if reset else 0), "crk_debug_vq_flags")
    return int(out[1]), int(out[2])


@pytest.mark.parametrize("case", ["normal", "tiny_codebook", "scaled_1e-4", "scaled_3e3", "duplicates", "near_ties", "zeros_and_padding",
                                  "small_K_384", "outliers", "ema_dead_codes", "ragged_K_100", "nan_and_inf_rows",
                                  "zero_rows_zero_codes"])
def test_vq_split_f16_search_equals_the_exact_fp32_search(case):
    """The default search (split-f16 on the matrix pipe + exact re-scoring of the undecided frames) against the exact
    fp32-MFMA search (crk_debug_vq_set_f16(0)): identical indices on 32 000 frames per case - random data at several
    scales, duplicated rows (exact ties -> lowest index), rows pulled 1e-7 apart (near ties), zero frames, a ragged frame
    count, a codebook that does not fill its tiles, activations with outliers.  The counters say how often the
    re-scoring paths ran: a fraction of a percent on random data, every frame that sits on a tie."""
    from crank_amd import _lib, ops

    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    N, K = 32000, 512
    x = torch.randn(N, 64, generator=g)
    w = torch.randn(K, 64, generator=g) * 0.5
    if case == "tiny_codebook":
        w = (torch.rand(K, 64, generator=g) * 2 - 1) / 512  # the reference's initial embedding (vqvae2.py:296-304)
    elif case == "scaled_1e-4":
        x, w = x * 1e-4, w * 1e-4
    elif case == "scaled_3e3":
        x, w = x * 3e3, w * 2e3
    elif case == "duplicates":
        w[100], w[300], w[301] = w[7], w[7], w[7]  # three copies: the full-scan path
        w[450] = w[20]                             # two copies: the two-candidate path
        x[:4000] = w[7] + 0.01 * x[:4000]
        x[4000:8000] = w[20] + 0.01 * x[4000:8000]
    elif case == "near_ties":
        w[200] = w[10] * (1 + 1e-7)
        w[201] = w[10] + 1e-7
        x[:8000] = w[10] + 0.05 * x[:8000]
    elif case == "zeros_and_padding":
        x[::3] = 0.0
        x, N = x[:31987], 31987
    elif case == "small_K_384":
        w, K = w[:384], 384
    elif case == "zero_rows_zero_codes":
        # an all-zero frame in front of two all-zero codes: every approximate distance of the pair is exactly 0, the search's
        # packed keys differ only by their tags (code 5 carries a smaller tag than code 2) - the exact tie goes to code 2
        x[::3] = 0.0
        w[5] = 0.0
        w[2] = 0.0
    elif case == "ema_dead_codes":  # quirk Q2: never-used codes end up at ~1e5 after the reference's EMA update
        w[::8] = torch.randn(64, 64, generator=g) * 1e5
    elif case == "ragged_K_100":  # the last code tile is partly padding
        w, K = w[:100], 100
    elif case == "nan_and_inf_rows":  # both searches pin such rows to code 0 (no distance compares below infinity)
        x[7] = float("nan")
        x[11, 3] = float("inf")
        x[13, 60] = float("-inf")
    elif case == "outliers":
        x[:, 5] *= 300.0
        x[::7, 40] = 1e-12
    xc, wc = x.cuda().contiguous(), w.cuda().contiguous()
    try:
        _lib.check(L.crk_debug_vq_set_f16(0), "set")
        e0, q0, i0 = ops.vq_apply(xc.view(1, N, 64), wc)
        _lib.check(L.crk_debug_vq_set_f16(1), "set")
        _vq_flags(reset=True)
        e1, q1, i1 = ops.vq_apply(xc.view(1, N, 64), wc)
        torch.cuda.synchronize()
        two, full = _vq_flags()
        # ... and with the codebook image prepared once (ops.vq_image_build) instead of derived by every workgroup: the same
        # decisions on the same frames (the re-scoring counters too), identical outputs
        img = torch.empty(ops.vq_image_bytes(K, 64), device="cuda", dtype=torch.uint8)
        ops.vq_image_build([wc], [img])
        _vq_flags(reset=True)
        e2, q2, i2 = ops.vq_apply(xc.view(1, N, 64), wc, image=img)
        torch.cuda.synchronize()
        assert _vq_flags() == (two, full)
        assert torch.equal(i1, i2) and torch.equal(e1, e2)
        assert torch.equal(torch.isnan(q1), torch.isnan(q2)) and torch.equal(torch.nan_to_num(q1), torch.nan_to_num(q2))
    finally:
        L.crk_debug_vq_set_f16(1)
    bad = (i0 != i1).nonzero()
    print(f"[vq f16 {case}] re-scored frames: two-candidate {two}, full scan {full} of {N}")
    assert bad.numel() == 0, (case, bad[:5].tolist(), i0.reshape(-1)[bad[:5, 1]].tolist(), i1.reshape(-1)[bad[:5, 1]].tolist())
    assert torch.equal(e0, e1)
    assert torch.equal(torch.isnan(q0), torch.isnan(q1)) and torch.equal(torch.nan_to_num(q0), torch.nan_to_num(q1))
    if case in ("normal", "tiny_codebook", "scaled_1e-4", "scaled_3e3", "small_K_384", "ema_dead_codes"):
        assert two + full < 0.02 * N, (two, full)  # the fast path decides nearly every frame of random data
    if case == "duplicates":
        assert full >= 3000 and two >= 3000, (two, full)
    if case == "zero_rows_zero_codes":
        assert (i1.reshape(-1)[::3] == 2).all() and two >= N // 3, two


def test_codebook_image_follows_every_write_to_the_codebook():
    """The generator keeps one prepared image per EMA codebook (VQVAE2.refresh_images) and counts the codebooks' states: two
    generators from one state dict, one with images and one without (ops.VQ_IMAGE off while its quantizers decide), must
    agree bit for bit on indices and codebooks through training forwards (every one blends the codebooks), a checkpoint load,
    an in-place write announced by touch(), and an optimizer step (which must NOT cost a rebuild)."""
    from crank_amd import ops
    from crank_amd.bin.train import get_model, get_optimizer
    from crank_amd.utils import load_yaml

    conf = load_yaml(None, batch_size=4, batch_len=200)
    torch.manual_seed(3)
    ga = get_model(conf, 4, "cuda")["G"]
    ops.VQ_IMAGE = False
    try:
        gb = get_model(conf, 4, "cuda")["G"]
        gb.load_state_dict(ga.state_dict())
        from crank_amd.synthetic import make_batch

        batch = make_batch(4, 200, 4, seed=5, device="cuda")
        x, dec_h = batch["in_feats"], torch.cat([batch["lcf0"], batch["uv"]], -1)
        h = batch["org_h"].clone()
        h[:, :] = h[:, 0:1]
        with torch.no_grad():
            gb.train()(x, None, dec_h, spkrvec=h)  # (its quantizers settle on "no image" now)
        assert all(q._img is False for q in gb.quantizers)
    finally:
        ops.VQ_IMAGE = True
    gb.load_state_dict(ga.state_dict())
    ga.train()

    def both():
        with torch.no_grad():
            oa, ob = ga(x, None, dec_h, spkrvec=h), gb(x, None, dec_h, spkrvec=h)
        for qa, qb in zip(oa["qidx"], ob["qidx"]):
            assert torch.equal(qa, qb)
        assert torch.equal(oa["decoded"], ob["decoded"])
        for qa, qb in zip(ga.quantizers, gb.quantizers):
            assert torch.equal(qa.weight, qb.weight)

    for _ in range(3):
        both()
    assert all(q._img is not None and q._img is not False for q in ga.quantizers)
    ep = ga.codebook_epoch
    sd = ga.state_dict()
    sd["quantizers.0.embedding.weight"] = torch.randn_like(sd["quantizers.0.embedding.weight"]) * 0.3
    ga.load_state_dict(sd); gb.load_state_dict(sd)
    assert ga.codebook_epoch > ep
    both()
    with torch.no_grad():
        ga.quantizers[1].weight.mul_(1.5); gb.quantizers[1].weight.mul_(1.5)
    ga.touch(); gb.touch()
    both()
    ga.eval(); gb.eval()
    both()
    ep = ga.codebook_epoch
    opt = get_optimizer(conf, {"G": ga})["G"]
    opt.step()  # zero gradients: the EMA codebooks stay as they are, and so do their images
    assert ga.codebook_epoch == ep
    both()
    both()


def test_vq_ema_matches_oracle_at_full_size():
    from crank_amd import ops
    from oracle.modules import vq_ema_update

    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(16, 500, 64, generator=g)
    w = torch.randn(512, 64, generator=g) * 0.5
    ema_size, ema_w = torch.rand(512, generator=g) * 5, torch.randn(64, 512, generator=g)
    xc, wc = x.cuda(), w.clone().cuda()
    _, _, idx = ops.vq_apply(xc, wc)
    s_ref, w_ref, cb_ref = vq_ema_update(x, idx.cpu(), ema_size.clone(), ema_w.clone())
    es, ew = ema_size.clone().cuda(), ema_w.clone().cuda()
    ops.vq_ema_update(xc, idx, es, ew, wc, 0.99, 1e-5)
    np.testing.assert_allclose(es.cpu().numpy(), s_ref.numpy(), rtol=2e-5)
    np.testing.assert_allclose(ew.cpu().numpy(), w_ref.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(wc.cpu().numpy(), cb_ref.numpy(), rtol=1e-4, atol=1e-6)
    # determinism: integer statistics make the update bitwise reproducible
    es2, ew2, wc2 = ema_size.clone().cuda(), ema_w.clone().cuda(), w.clone().cuda()
    ops.vq_ema_update(xc, idx, es2, ew2, wc2, 0.99, 1e-5)
    assert torch.equal(es2, es) and torch.equal(ew2, ew) and torch.equal(wc2, wc)


def test_vq_fused_input_sum_and_commit_loss_equal_the_composed_ops():
    """crk_vq_forward_fused (x + add and the commitment partials inside the search kernel) against the separate
    launches: same indices, code vectors, straight-through values and input sum bit for bit; the loss to summation order;
    the same gradients for both addends."""
    from crank_amd import ops

    torch.manual_seed(3)
    B, T, D, K = 5, 333, 64, 512
    cb = torch.randn(K, D, device="cuda") * 0.3
    xh, ah = torch.randn(B, T, D, device="cuda"), 0.5 * torch.randn(B, T, D, device="cuda")
    mask = torch.rand(B, T, device="cuda") > 0.25
    w = torch.randn(B, T, D, device="cuda")

    def run(fused):
        x, a = xh.clone().requires_grad_(True), ah.clone().requires_grad_(True)
        if fused:
            e, qx, idx, commit, xin = ops.vq_commit_apply(x, cb, mask, add=a)
        else:
            xin = x + a
            e, qx, idx, commit = ops.vq_commit_apply(xin, cb, mask)
        ((qx * w).sum() + 0.25 * commit + 0.1 * (xin ** 2).sum()).backward()
        return e, qx, idx, commit.item(), xin.detach(), x.grad, a.grad

    f, c = run(True), run(False)
    for i in (0, 1, 2, 4):
        assert torch.equal(f[i], c[i]), i
    np.testing.assert_allclose(f[3], c[3], rtol=2e-6)
    ref = ((c[4] - c[0]) ** 2)[mask].mean().item()
    np.testing.assert_allclose(f[3], ref, rtol=1e-5)
    for i in (5, 6):
        np.testing.assert_allclose(f[i].cpu().numpy(), c[i].cpu().numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(f[5], f[6])
    # without the loss: indices only path, no mask
    with torch.no_grad():
        r = ops.vq_apply(xh, cb, add=ah)
        r2 = ops.vq_apply(xh + ah, cb)
    assert torch.equal(r[2], r2[2]) and torch.equal(r[1], r2[1]) and torch.equal(r[3], xh + ah)


@pytest.mark.parametrize("K", [512, 100, 33])
def test_ema_blend_that_leaves_the_search_image_equals_blend_then_build(K):
    """crk_vq_ema_blend_image_multi (the blend and the codebook's search image in one launch) against
    crk_vq_ema_blend_multi followed by crk_vq_image_build_multi: the same ema_w, codebook and image, byte for byte - for a
    full codebook and for ones whose last 32-code tiles are partly / wholly padding."""
    from crank_amd import ops

    torch.manual_seed(11)
    D = 64
    nq = 3
    sums = [(torch.randn(D, K, device="cuda") * 2.0 ** 28 * 7).to(torch.int64) for _ in range(nq)]
    size0 = [torch.rand(K, device="cuda") * 50 + 0.01 for _ in range(nq)]
    w0 = [torch.randn(D, K, device="cuda") * 30 for _ in range(nq)]
    out = []
    for fused in (False, True):
        size, w = [t.clone() for t in size0], [t.clone() for t in w0]
        cb = [torch.zeros(K, D, device="cuda") for _ in range(nq)]
        img = [torch.full((ops.vq_image_bytes(K, D),), 0x5A, device="cuda", dtype=torch.uint8) for _ in range(nq)]
        done = ops.vq_ema_blend_multi(sums, size, w, cb, [D] * nq, [K] * nq, 0.99, images=img if fused else None)
        assert done is fused
        if not fused:
            ops.vq_image_build(cb, img)
        out.append((w, cb, img))
    for a, b in zip(out[0], out[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_vq_input_sum_with_the_prepared_image_and_summed_in_place():
    """With a prepared codebook image the search kernel stores x + add where it forms it (under the search) instead of in
    its gather epilogue: the same sum, outputs and loss as without the image, bit for bit - and a caller of the C ABI that
    sums IN PLACE (xsum == x, or == add) still gets the outputs of the inputs it passed (the epilogue reads them again)."""
    from crank_amd import _lib, ops
    from crank_amd.ops import ptr, stream_ptr

    torch.manual_seed(5)
    B, T, D, K = 3, 401, 64, 512
    cb = torch.randn(K, D, device="cuda") * 0.3
    x, a = torch.randn(B, T, D, device="cuda"), 0.5 * torch.randn(B, T, D, device="cuda")
    mask = torch.rand(B, T, device="cuda") > 0.25
    img = torch.empty(ops.vq_image_bytes(K, D), device="cuda", dtype=torch.uint8)
    ops.vq_image_build([cb], [img])
    with torch.no_grad():
        r0 = ops.vq_commit_apply(x, cb, mask, add=a)
        r1 = ops.vq_commit_apply(x, cb, mask, add=a, image=img)
    for i in (0, 1, 2, 4):
        assert torch.equal(r0[i], r1[i]), i
    assert r0[3].item() == r1[3].item()
    assert torch.equal(r1[4], x + a)
    L = _lib.lib()
    for alias in ("x", "add"):
        xc, ac = x.clone(), a.clone()
        tgt = xc if alias == "x" else ac
        idx = torch.empty(B, T, device="cuda", dtype=torch.int64)
        e, qx = torch.empty_like(x), torch.empty_like(x)
        rc = L.crk_vq_forward_fused(ptr(xc), D, ptr(ac), D, ptr(tgt), D, ptr(cb), B * T, D, K, ptr(idx), ptr(e), D, ptr(qx), D,
                                    None, None, None, ptr(img), stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(idx, r0[2]) and torch.equal(e, r0[0]) and torch.equal(qx, r0[1]) and torch.equal(tgt, x + a), alias


@pytest.mark.parametrize("with_add", [False, True])
@pytest.mark.parametrize("consumers", ["none", "x", "qx", "both"])
@pytest.mark.parametrize("wide", [False, True])
def test_vq_commit_aliases_join_the_gradients_of_second_consumers_bitwise(with_add, consumers, wide):
    """_VQCommitFn(alias=True) hands out x and qx a second time; a consumer of an alias sends its gradient into the op's
    ONE backward launch (crk_vq_commit_bwd).  Against the same graph built on the tensors themselves (autograd adds the
    gradients in launches of its own): x.grad, add.grad identical to the bit - with x and qx contiguous or column slices of
    wider buffers (the strided gradients of the real step), with and without an input sum, for every set of consumers."""
    from crank_amd import ops

    torch.manual_seed(11)
    B, T, D, K = 3, 250, 64, 512
    cb = torch.randn(K, D, device="cuda") * 0.3
    xw = torch.randn(B, T, 2 * D, device="cuda")
    ah = 0.5 * torch.randn(B, T, D, device="cuda")
    mask = torch.rand(B, T, device="cuda") > 0.25
    w1, w2 = torch.randn(B, T, D, device="cuda"), torch.randn(B, T, 2 * D, device="cuda")
    w3 = torch.randn(B, T, 2 * D, device="cuda")

    def run(alias):
        xbuf = xw.clone().requires_grad_(True)
        x = xbuf[..., D:] if wide else xbuf[..., :D].contiguous()
        a = ah.clone().requires_grad_(True) if with_add else None
        qbuf = torch.zeros(B, T, 2 * D, device="cuda") if wide else None
        r = ops.vq_commit_apply(x, cb, mask, qx_out=(qbuf, D) if wide else None, add=a, alias=alias)
        e, qx, idx, commit = r[:4]
        x2, qx2 = (r[-2], r[-1]) if alias else (x, qx)
        loss = (qx * w1).sum() + 0.25 * commit
        if consumers in ("x", "both"):  # a strided gradient, as the speaker-adversarial net's column slice is
            loss = loss + (torch.cat([x2, w1], dim=-1) * w2).sum()
        if consumers in ("qx", "both"):
            loss = loss + (torch.cat([w1, qx2], dim=-1) * w3).sum()
        loss.backward()
        return xbuf.grad, (a.grad if with_add else None), idx

    j, s = run(True), run(False)
    assert torch.equal(j[2], s[2])
    assert j[0].abs().max() > 0
    assert torch.equal(j[0], s[0]), float((j[0] - s[0]).abs().max())
    if with_add:
        assert torch.equal(j[1], s[1]), float((j[1] - s[1]).abs().max())


def test_embedding_lookup_reads_the_first_label_of_an_utterance_without_a_filled_copy():
    """concat_embed with the stride-0 view ``h[:, 0:1].expand(-1, T)`` of the batch's labels (what the trainers hand over
    on the GPU: crk_concat_embed_run / crk_embed_bwd_run with run = T) against the filled contiguous copy the reference
    makes (basetrainer.py:303-308), labels with -100 pads behind each utterance: same rows, same table gradient, same
    gradients of the concatenated tensors, bit for bit; and against plain indexing."""
    from crank_amd import ops

    torch.manual_seed(5)
    B, T, S, E = 6, 211, 9, 32
    h = torch.full((B, T), -100, dtype=torch.long, device="cuda")
    for b in range(B):
        h[b, : 40 + 17 * b] = (3 * b + 1) % S
    a, c = torch.randn(B, T, 1, device="cuda"), torch.randn(B, T, 1, device="cuda")
    w = torch.randn(B, T, 2 + E, device="cuda")

    class Owner:
        skip_param_grads = False
        grads_clean = True

    def run(view):
        o = Owner()
        o.flat = torch.randn(S * E + 7, device="cuda")
        torch.manual_seed(6)
        o.flat.copy_(torch.randn(S * E + 7, device="cuda"))
        o.grad_flat = torch.zeros_like(o.flat)
        table = o.flat[7:].view(S, E)
        idx = h[:, 0:1].expand(-1, T)
        idx = idx if view else idx.contiguous()
        aa, cc = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
        out = ops.concat_embed(aa, cc, table, idx, o, 7, o.flat)
        (out * w).sum().backward()
        return out.detach(), o.grad_flat.clone(), aa.grad, cc.grad, table

    v, f = run(True), run(False)
    for i in range(4):
        assert torch.equal(v[i], f[i]), i
    assert v[1].abs().max() > 0
    ref = torch.cat([a, c, v[4][h[:, 0]][:, None, :].expand(-1, T, -1)], dim=-1)
    assert torch.equal(v[0], ref)


# ------------------------------------------------------------------ losses
def test_feature_losses_vs_reference_values_and_grads():
    from crank_amd.net.module.loss import CustomFeatureLoss

    fx = golden("losses.npz")
    y, mask = cu(fx["y"]), cu(fx["mask"])
    sp = {"fft_sizes": [64, 128], "win_sizes": [64, 128], "hop_sizes": [16, 32], "logratio": 0}
    for causal in [False, True]:
        for cs in ([0] if not causal else [-8, -2, 0, 2, 8]):
            for lt in ["l1", "mse", "stft"]:
                crit = CustomFeatureLoss(loss_type=lt, causal=causal, stft_params=sp)
                x = cu(fx["x"]).requires_grad_(True)
                v = crit(x, y, mask=None if lt == "stft" else mask, causal_size=cs)
                v.backward()
                tag = f"{lt}_c{int(causal)}_cs{cs}"
                np.testing.assert_allclose(v.item(), float(fx[f"val_{tag}"]), rtol=2e-5, err_msg=tag)
                gref = fx[f"grad_{tag}"]
                np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=1e-3, atol=2e-5 * np.abs(gref).max(),
                                           err_msg=tag)
    for lt in ["l1", "mse"]:
        x = cu(fx["x"]).requires_grad_(True)
        v = CustomFeatureLoss(loss_type=lt)(x, y)
        v.backward()
        np.testing.assert_allclose(v.item(), float(fx[f"val_{lt}_nomask"]), rtol=2e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), fx[f"grad_{lt}_nomask"], rtol=1e-4, atol=1e-9)


def test_stft_loss_variants_vs_reference():
    from crank_amd.net.module.loss import MultiSizeSTFTLoss, STFTLoss

    fx = golden("losses.npz")
    y = cu(fx["y"])
    x = cu(fx["x"]).requires_grad_(True)
    v = STFTLoss(fft_size=32, win_size=20, hop_size=10, logratio=0.3)(x, y)
    v.backward()
    np.testing.assert_allclose(v.item(), float(fx["val_stftloss_direct"]), rtol=5e-5)
    g = fx["grad_stftloss_direct"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), g, rtol=2e-3, atol=5e-5 * np.abs(g).max())
    x = cu(fx["x"]).requires_grad_(True)
    v = MultiSizeSTFTLoss(fft_sizes=[32, 64], win_sizes=[32, 64], hop_sizes=[8, 16], logratio=0.25)(x, y)
    v.backward()
    np.testing.assert_allclose(v.item(), float(fx["val_ms_log"]), rtol=5e-5)
    g = fx["grad_ms_log"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), g, rtol=2e-3, atol=5e-5 * np.abs(g).max())


def _torch_recon(x, y, mask, resolutions, logratio):
    """(L1, MSE, STFT loss) with torch ops on the CPU: loss.py:30-47 (masked_select + mean) and :50-114 (torch.stft)."""
    x, y = x.double(), y.double()
    sel = mask.reshape(mask.shape[0], mask.shape[1], 1).expand_as(x) if mask is not None else torch.ones_like(x, dtype=torch.bool)
    l1 = (x - y).abs()[sel].mean()
    mse = ((x - y) ** 2)[sel].mean()
    B, T, D = x.shape
    total = 0.0
    for n_fft, hop, win in resolutions:
        w = torch.hann_window(win, dtype=torch.float64)

        def mag(v):
            s = torch.stft(v.permute(0, 2, 1).reshape(-1, T), n_fft, hop, win, w, return_complex=True)
            return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=1e-7))
        mx, my = mag(x), mag(y)
        total = total + (1 - logratio) * (mx - my).abs().mean() + logratio * (mx.log() - my.log()).abs().mean()
    return l1, mse, total / len(resolutions)


@pytest.mark.parametrize("case", [
    dict(B=3, T=500, D=80, res=[(64, 64, 16), (128, 128, 32)], lr=0.0),   # the step's geometry (quirk Q1)
    dict(B=2, T=257, D=20, res=[(32, 19, 8)], lr=0.3),                    # odd T, odd hop, log term
    dict(B=2, T=256, D=12, res=[(64, 64, 16), (32, 32, 9)], lr=0.2),      # a frame centred on T, odd window
    dict(B=1, T=130, D=3, res=[(128, 70, 64)], lr=0.0),                   # 64-tap tile, D not a multiple of 4
    dict(B=2, T=96, D=8, res=[(16, 11, 8), (32, 40, 32), (8, 7, 4)], lr=0.0, sliced=True),  # three resolutions, row stride > D
])
def test_recon_loss_fused_vs_torch_and_dense_path(case, monkeypatch):
    """The one-launch reconstruction losses (compact STFT gradient, no atomics) against torch.stft on the CPU and
    against the dense / atomic path they replace; every reflect-padding corner is in the cases."""
    from crank_amd import config

    from crank_amd import ops

    B, T, D, res, lr = case["B"], case["T"], case["D"], case["res"], case["lr"]
    gen = torch.Generator().manual_seed(5)
    xh = torch.randn(B, T, D + (4 if case.get("sliced") else 0), generator=gen)[..., :D]
    yh = xh + 0.3 * torch.randn(B, T, D, generator=gen)
    mh = torch.rand(B, T, generator=gen) > 0.2
    wts = (2.0, 0.5, 1.0)
    windows = [torch.hann_window(w, dtype=torch.float32, device="cuda") for _, _, w in res]
    assert ops.recon_supported(T, res)

    def run(dense):
        monkeypatch.setattr(config.cfg, "recon_dense", bool(dense))
        if case.get("sliced"):  # x is a column slice of a wider tensor
            full = torch.randn(B, T, D + 4, device="cuda")
            full[..., :D] = xh.cuda()
            leaf = full.requires_grad_(True)
            xg = leaf[..., :D]
        else:
            leaf = xh.cuda().contiguous().requires_grad_(True)
            xg = leaf
        vals = ops.recon_loss(xg, yh.cuda(), mh.cuda(), res, windows, lr)
        sum(w * v for w, v in zip(wts, vals)).backward()
        return [v.item() for v in vals], leaf.grad[..., :D].cpu().numpy()

    vf, gf = run(False)
    vd, gd = run(True)
    xr = xh.clone().double().requires_grad_(True)
    ref = _torch_recon(xr, yh, mh, res, lr)
    sum(w * v for w, v in zip(wts, ref)).backward()
    gr = xr.grad.numpy()
    np.testing.assert_allclose(vf, [v.item() for v in ref], rtol=3e-5)
    np.testing.assert_allclose(vf, vd, rtol=2e-6)
    np.testing.assert_allclose(gf, gr, rtol=2e-3, atol=3e-5 * np.abs(gr).max())
    np.testing.assert_allclose(gf, gd, rtol=2e-3, atol=2e-6 * np.abs(gr).max())
    # one term differentiated alone, and none at all
    for pick in range(3):
        leaf = xh.cuda().contiguous().requires_grad_(True)
        ops.recon_loss(leaf, yh.cuda(), mh.cuda(), res, windows, lr)[pick].backward()
        xr = xh.clone().double().requires_grad_(True)
        _torch_recon(xr, yh, mh, res, lr)[pick].backward()
        g = xr.grad.numpy()
        np.testing.assert_allclose(leaf.grad.cpu().numpy(), g, rtol=2e-3, atol=3e-5 * np.abs(g).max(), err_msg=str(pick))
    with torch.no_grad():
        v = ops.recon_loss(xh.cuda().contiguous(), yh.cuda(), None, res, windows, lr)
    rn = _torch_recon(xh, yh, None, res, lr)
    np.testing.assert_allclose([t.item() for t in v], [t.item() for t in rn], rtol=3e-5)
    # bitwise repeatable (no atomics)
    a = run(False)
    assert a[0] == vf and np.array_equal(a[1], gf)


def test_cross_entropy_ignore_index_vs_reference():
    from crank_amd.net.module.loss import CrossEntropyLoss

    fx = golden("losses.npz")
    logits = cu(fx["ce_logits"]).requires_grad_(True)
    v = CrossEntropyLoss(-100)(logits, cu(fx["ce_target"]))
    (3.0 * v).backward()
    np.testing.assert_allclose(v.item(), float(fx["ce_val"]), rtol=1e-5)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), 3.0 * fx["ce_grad"], rtol=1e-4, atol=1e-9)


def test_lsgan_constant_target_and_empty_mask():
    from crank_amd import ops

    x = torch.randn(4, 50, 1, device="cuda", requires_grad=True)
    mask = torch.rand(4, 50, 1, device="cuda") > 0.3
    v = ops.masked_mean_loss(x, None, mask, "mse", yconst=1.0)
    ref = ((x.detach()[mask] - 1) ** 2).mean()
    np.testing.assert_allclose(v.item(), ref.item(), rtol=1e-5)
    v.backward()
    gref = torch.where(mask, 2 * (x.detach() - 1) / mask.sum(), torch.zeros_like(x))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref.cpu().numpy(), rtol=1e-5, atol=1e-8)
    empty = ops.masked_mean_loss(x, None, torch.zeros_like(mask), "mse", yconst=1.0)
    assert torch.isnan(empty)  # mean of an empty selection is NaN in the reference too


# ------------------------------------------------------------------ glue
def test_adam_matches_torch_adam():
    from crank_amd import ops

    torch.manual_seed(0)
    p0 = torch.randn(10007)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-4)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr, step = torch.tensor([2e-4], device="cuda"), torch.zeros(1, device="cuda")
    for i in range(5):
        g = torch.randn(10007) * (0.1 + i)
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.cuda(), m, v, lr, step)
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    assert step.item() == 5


def test_radam_matches_the_published_update():
    """crank/net/trainer/utils.py:44-45 (torch_optimizer.RAdam(lr): absent package, published update restated in
    oracle/optim.py::RAdam, itself held against torch.optim.RAdam on CPU).  12 steps: the momentum-only regime (N_sma < 5
    up to step 5) and the rectified one, the switch at the same step as the oracle's."""
    from crank_amd import ops
    from oracle.optim import RAdam

    torch.manual_seed(0)
    p0 = torch.randn(10007) * 1e-3  # small parameters: an fp32 ulp of p is ~1e-5 of a step's movement, the movement is what is compared
    ref = torch.nn.Parameter(p0.clone())
    opt = RAdam([ref], lr=2e-4)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr, step = torch.tensor([2e-4], device="cuda"), torch.zeros(1, device="cuda")
    for i in range(12):
        g = torch.randn(10007) * (0.1 + i)
        ref.grad = g.clone()
        before = ref.detach().clone()
        opt.step()
        gd = g.cuda()
        pb = p.clone()
        ops.radam_step(p, gd, m, v, lr, step, clear_grads=True)
        assert not gd.any()  # consumed and cleared
        # the update itself (not only the parameter it is a 1e-4 part of)
        np.testing.assert_allclose((p - pb).cpu().numpy(), (ref.detach() - before).numpy(), rtol=2e-5, atol=1e-9, err_msg=f"step {i}")
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    # (moments of order 1 / 100 whose elements can cancel to ~0: absolute floors of a few ulps of that order)
    np.testing.assert_allclose(m.cpu().numpy(), opt.state[ref]["exp_avg"].numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(v.cpu().numpy(), opt.state[ref]["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-6)
    assert step.item() == 12


class _FlatStub:
    """What FlatAdam and its subclasses touch of a FlatModel."""

    def __init__(self, entries, n):
        self._entries = entries
        self.flat = torch.nn.Parameter(torch.zeros(n, device="cuda"))
        self.grad_flat = torch.zeros(n, device="cuda")
        self.grads_clean = True
        self.version = 1

    def touch(self, by_optimizer=False):
        self.version += 1


def test_lamb_matches_the_published_update_per_parameter_tensor():
    """crank/net/trainer/utils.py:46-47 (pytorch_lamb.Lamb(lr): absent package, published update restated in
    oracle/optim.py::Lamb).  The flat block's parameter tensors are its entries: a trust ratio each, weight norm clamped at
    10, ratio 1 for an all-zero tensor, a tensor without gradient (an EMA codebook) left bit for bit alone, a hole between
    two entries a tensor of its own; tiles never cross a tensor."""
    from crank_amd import ops
    from crank_amd.net.trainer.utils import FlatLamb
    from oracle.optim import Lamb

    torch.manual_seed(1)
    shapes = [("a.weight_v", (128, 64, 5)), ("a.weight_g", (128, 1, 1)), ("a.bias", (128,)), ("big", (300, 70)),
              ("zeros", (64, 3)), ("codebook", (512, 64)), ("tail", (7,))]
    entries, off = [], 0
    for k, shp in shapes:
        entries.append((k, off, shp))
        off += int(np.prod(shp))
        if k == "a.bias":
            off += 5  # a hole in the block
    n = off + 3
    model = _FlatStub(entries, n)
    vals = {k: torch.randn(shp) * (3.0 if k == "big" else 0.3) for k, shp in shapes}
    vals["zeros"].zero_()
    refs = {k: torch.nn.Parameter(vals[k].clone()) for k, _ in shapes}
    for k, o, shp in entries:
        model.flat.data[o: o + vals[k].numel()] = vals[k].reshape(-1).cuda()
    opt = FlatLamb(model, 1e-3)
    ref_opt = Lamb(list(refs.values()), lr=1e-3)
    tile = ops.lamb_tile()
    tiles, tensors = opt.tiles.cpu().numpy(), opt.tensors.cpu().numpy()
    covered = np.zeros(n, dtype=np.int32)
    for o, ln, t, _ in tiles:
        assert 0 < ln <= tile
        so, sn = opt.tensor_spans[t]
        assert so <= o and o + ln <= so + sn
        covered[o: o + ln] += 1
    assert (covered == 1).all()
    assert [int(f) for f, _ in tensors] == list(np.cumsum([0] + [int(c) for _, c in tensors[:-1]]))
    assert len(opt.tensor_spans) == len(shapes) + 2  # the hole and the block's last 3 elements
    assert float(refs["big"].detach().norm()) > 10  # the clamp is exercised
    code0 = model.flat.data[entries[5][1]: entries[5][1] + 512 * 64].clone()
    for i in range(6):
        for k, o, shp in entries:
            if k == "codebook":
                continue  # no gradient: the reference's optimizer skips it, the flat one sees zeros
            g = torch.randn(shp) * (0.05 + 0.2 * i)
            refs[k].grad = g.clone()
            model.grad_flat[o: o + g.numel()] = g.reshape(-1).cuda()
        ref_opt.step()
        opt.step()
        assert not model.grad_flat.any()
        for s, (k, o, shp) in enumerate(entries):
            if k == "codebook":
                continue
            cnt = int(np.prod(shp))
            # the three factors of the movement one by one (the movement itself is ~1e-3 of a weight: an fp32 ulp of the
            # weight is 1e-4 of it), then the weights to an ulp
            st = ref_opt.state[refs[k]]
            u_ref = (st["exp_avg"] / st["exp_avg_sq"].sqrt().add(1e-6)).reshape(-1).numpy()
            np.testing.assert_allclose(opt.upd[o: o + cnt].cpu().numpy(), u_ref, rtol=1e-5, atol=1e-6, err_msg=f"u of {k} step {i}")
            t = [j for j, sp in enumerate(opt.tensor_spans) if sp[0] == o][0]
            np.testing.assert_allclose(opt.trust_ratio[t].item(), st["trust_ratio"], rtol=2e-5, err_msg=k)
            np.testing.assert_allclose(model.flat.data[o: o + cnt].cpu().numpy(), refs[k].detach().reshape(-1).numpy(),
                                       rtol=3e-7, atol=1e-9, err_msg=f"{k} step {i}")
    assert torch.equal(model.flat.data[entries[5][1]: entries[5][1] + 512 * 64], code0)
    assert opt.step_dev.item() == 6
    # a saved state goes back in
    sd = opt.state_dict()
    opt2 = FlatLamb(model, 1e-3)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)


@pytest.mark.parametrize("optim_type", ["radam", "lamb"])
def test_trainer_step_with_the_other_optimizers_of_the_factory(optim_type):
    """crank/net/trainer/utils.py:40-50 inside the step: the vqvae scenario of the goldens with optim.*.type = radam / lamb,
    product trainer (bf16x3) against the CPU oracle trainer with oracle/optim.py's optimizers: every loss of 7 steps (RAdam
    changes regime behind step 5) and every parameter tensor's MOVEMENT over the run."""
    from crank_amd import ops
    from crank_amd.net.trainer.utils import FlatLamb, FlatRAdam
    from tests.helpers import run_golden_case
    from tests.test_gpu_step import _hip_factories, _oracle_factories

    torch.set_num_threads(8)
    steps = 7
    lo, mo, _, _, _ = run_golden_case("vqvae", *_oracle_factories(), optim_type=optim_type, steps=steps)
    ops.set_precision("bf16x3")
    try:
        lh, mh, th, _, _ = run_golden_case("vqvae", *_hip_factories(), device="cuda", optim_type=optim_type, steps=steps)
    finally:
        ops.set_precision("bf16")
    assert all(isinstance(o, FlatLamb if optim_type == "lamb" else FlatRAdam) for o in th.optimizer.values())
    for s in range(steps):
        for k, v in lo[s].items():
            if v:
                assert abs(lh[s][k] - v) <= 2e-3 * abs(v) + 1e-5, (s, k, lh[s][k], v)
    from tests.helpers import initial_state

    init = initial_state(mo)
    worst = 0.0
    for m in mo:
        so, sh = mo[m].state_dict(), mh[m].state_dict()
        for k in so:
            do = (so[k].float() - init[m][k].float()).reshape(-1)
            dh = (sh[k].float().cpu() - init[m][k].float()).reshape(-1)
            if float(do.norm()) == 0.0:
                assert float(dh.norm()) == 0.0, (m, k)
                continue
            err = float((dh - do).norm() / do.norm())
            worst = max(worst, err)
            assert err < 0.05, (m, k, err)
    print(f"{optim_type}: worst relative error of a tensor's movement {worst:.3e}")


def test_gradient_reversal_and_steplr_goldens():
    from crank_amd.net.trainer.utils import StepLR

    fx = golden("misc.npz")

    class Opt:
        base_lr = 2e-4
        param_groups = [{"lr": 2e-4}]

        def set_lr(self, lr):
            self.param_groups[0]["lr"] = lr

    o = Opt()
    s = StepLR(o, 200000, 0.5)
    for st, lr in zip(fx["steplr_steps"], fx["steplr_lr"]):
        s.step(int(st))
        assert abs(o.param_groups[0]["lr"] - float(lr)) < 1e-12
    # GRL: identity forward, -scale * g backward, through the SPKRADV stack's dx_scale
    from crank_amd import ops
    from crank_amd.net.module.spkradv import SpeakerAdversarialNetwork
    from crank_amd.utils import load_yaml

    ops.set_precision("bf16x3")
    conf = load_yaml(None)
    net = SpeakerAdversarialNetwork(conf, 4)
    x0 = torch.randn(2, 40, 64, device="cuda", requires_grad=True)
    x1 = torch.randn(2, 40, 64, device="cuda", requires_grad=True)
    net.forward([x0, x1]).sum().backward()
    g_rev = x0.grad.clone()
    net.scale = -net.scale  # flipping the sign flips the gradient, nothing else
    x0.grad = None
    net.forward([x0, x1]).sum().backward()
    np.testing.assert_allclose(x0.grad.cpu().numpy(), -g_rev.cpu().numpy(), rtol=1e-6, atol=1e-9)
    x0.grad = None
    net.forward([x0, x1], detach=True).sum().backward()
    assert x0.grad is None
    ops.set_precision("bf16")


def test_logmel_layer_vs_oracle_and_reference_stft():
    from crank_amd.net.module.mlfb import LogMelFilterBankLayer
    from oracle.modules import OracleLogMel

    fx = golden("stft_layer.npz")
    fs = int(fx["fs"])
    wav = torch.from_numpy(fx["wav"])[None]
    kw = dict(fs=fs, hop_size=128, fft_size=1024, win_length=1024, window="hann", center=False, n_mels=80,
              fmin=80, fmax=7600)
    orac = OracleLogMel(**kw)
    # the oracle's STFT magnitudes equal the reference STFTLayer's (golden, every 8th frame)
    s = orac.stft(wav)
    amp = torch.sqrt(s[..., 0] ** 2 + s[..., 1] ** 2).numpy()[:, ::8]
    np.testing.assert_allclose(amp, fx["amp_center0"], rtol=1e-4, atol=1e-5)
    ref = orac(wav).numpy()
    got = LogMelFilterBankLayer(**kw)(wav.cuda()).cpu().numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-3)  # log10 domain; reference tests use decimal=3

    class Sc:
        mean_, var_ = fx["scaler_mean"], fx["scaler_var"]

    got = LogMelFilterBankLayer(**kw, scaler=Sc)(wav.cuda()).cpu().numpy()
    ref = OracleLogMel(**kw, scaler=Sc)(wav).numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-3)


def test_offline_logmel_extraction_centered_reflect():
    """The stage-2 extraction (feature.py:126-145 -> logmelfilterbank): center=True with reflect padding.
    The oracle's centred STFT is pinned to the reference STFTLayer(center=True) fixture; ragged lengths
    exercise both mirrored edges (shortest: barely longer than the padding)."""
    from crank_amd.net.module.mlfb import logmelfilterbank
    from oracle.modules import OracleLogMel

    fx = golden("stft_layer.npz")
    fs = int(fx["fs"])
    kw = dict(fs=fs, hop_size=128, fft_size=1024, win_length=1024, window="hann", center=True, n_mels=80, fmin=80, fmax=7600)
    orac = OracleLogMel(**kw)
    wav = torch.from_numpy(fx["wav"])[None]
    s = orac.stft(wav)
    amp = torch.sqrt(s[..., 0] ** 2 + s[..., 1] ** 2).numpy()[:, ::8]
    np.testing.assert_allclose(amp, fx["amp_center1"], rtol=1e-4, atol=1e-5)
    for n in (fx["wav"].shape[0], 9999, 1500, 513):
        x = fx["wav"][:n]
        ref = orac(torch.from_numpy(x)[None]).numpy()[0]
        got = logmelfilterbank(x, fs, fft_size=1024, hop_size=128, win_length=1024, window="hann", num_mels=80, fmin=80, fmax=7600)
        assert got.shape == ref.shape == (1 + n // 128, 80)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-3)
    with pytest.raises(ValueError):
        logmelfilterbank(fx["wav"][:512], fs, fft_size=1024, hop_size=128, num_mels=80, fmin=80, fmax=7600)


def test_fused_ema_of_all_quantizers_equals_the_per_quantizer_update_bitwise():
    """One reduce + one blend launch for every quantizer of a generator forward (crk_vq_ema_reduce_multi /
    crk_vq_ema_apply_multi) against crk_vq_ema_stats + crk_vq_ema_apply per quantizer: identical bits, twice in a row."""
    from crank_amd import ops

    torch.manual_seed(3)
    dims = [(64, 512), (32, 128)]
    N = 4000
    xs = [torch.randn(N, D, device="cuda") for D, _ in dims]
    idx = [torch.randint(0, K, (N,), device="cuda") for _, K in dims]
    state = [(torch.rand(K, device="cuda") * 5, torch.randn(D, K, device="cuda"), torch.zeros(K, D, device="cuda")) for D, K in dims]
    ref = [tuple(t.clone() for t in s) for s in state]
    for _ in range(2):
        counts = [torch.empty(K, device="cuda", dtype=torch.int32) for _, K in dims]
        sums = [torch.empty(D * K, device="cuda", dtype=torch.int64) for D, K in dims]
        parts = [ops.vq_ema_partial(x, i, D, K) for x, i, (D, K) in zip(xs, idx, dims)]
        ops.vq_ema_reduce_multi([p[0] for p in parts], [p[1] for p in parts], [d[0] for d in dims], [d[1] for d in dims], counts, sums)
        ops.vq_ema_apply_multi(counts, sums, [s[0] for s in state], [s[1] for s in state], [s[2] for s in state],
                               [d[0] for d in dims], [d[1] for d in dims], 0.99, 1e-5)
        for x, i, (D, K), r in zip(xs, idx, dims, ref):
            ops.vq_ema_update(x, i, r[0], r[1], r[2], 0.99, 1e-5)
        for s, r, c, i, (D, K) in zip(state, ref, counts, idx, dims):
            assert torch.equal(c.long(), torch.bincount(i, minlength=K))
            for a, b in zip(s, r):
                assert torch.equal(a, b)
