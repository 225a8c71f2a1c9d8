"""CPU: pin the oracle (oracle/modules.py) against the golden vectors generated from the
reference's own classes, and check the host-side pieces that need no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import REPO, golden


# ------------------------------------------------------------------ oracle vs goldens
def test_oracle_quantizer_matches_reference_quantizer():
    from oracle.modules import OracleQuantizer

    fx = golden("quantizer.npz")
    q = OracleQuantizer(64, 512, ema_flag=True, bdt_flag=True).train()
    q.embedding.weight.data.copy_(torch.from_numpy(fx["init_weight"]))
    q.ema_w.data.copy_(torch.from_numpy(fx["init_ema_w"]))
    q.ema_size.copy_(torch.from_numpy(fx["init_ema_size"]))
    for it in range(3):
        e, qx, idx = q(torch.from_numpy(fx[f"x{it}"]), use_ema=True)
        assert np.array_equal(idx.numpy(), fx[f"idx{it}"])
        np.testing.assert_allclose(e.detach().numpy(), fx[f"e{it}"], rtol=1e-6)
        np.testing.assert_allclose(qx.detach().numpy(), fx[f"qx{it}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(q.ema_size.numpy(), fx[f"ema_size{it}"], rtol=1e-6)
        np.testing.assert_allclose(q.ema_w.numpy(), fx[f"ema_w{it}"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(q.embedding.weight.detach().numpy(), fx[f"w{it}"], rtol=1e-5, atol=1e-7)
    e, qx, idx = q(torch.from_numpy(fx["x3"]), use_ema=False)
    assert np.array_equal(idx.numpy(), fx["idx3"])
    np.testing.assert_allclose(q.embedding.weight.detach().numpy(), fx["w3"], rtol=1e-6)
    q2 = OracleQuantizer(64, 512, ema_flag=False, bdt_flag=False).eval()
    q2.embedding.weight.data.copy_(torch.from_numpy(fx["tie_w"]))
    e, qx, idx = q2(torch.from_numpy(fx["tie_x"]))
    assert np.array_equal(idx.numpy(), fx["tie_idx"])


def test_oracle_vq_matches_reference_at_benchmark_size():
    """quantizer_full.npz (32 000 frames through the reference Quantizer): the oracle's argmin gives every index."""
    from oracle.modules import vq_nearest

    fx = golden("quantizer_full.npz")
    B, D, T = [int(v) for v in fx["x_shape_BDT"]]
    x = np.random.RandomState(int(fx["x_seed"])).standard_normal((B, D, T)).astype(np.float32)
    idx = vq_nearest(torch.from_numpy(x.transpose(0, 2, 1).reshape(-1, D).copy()), torch.from_numpy(fx["codebook"]))
    assert np.array_equal(idx.view(B, T).numpy(), fx["idx"].astype(np.int64))


def test_oracle_losses_match_reference_losses():
    from oracle.modules import OracleFeatureLoss, multi_stft_loss, stft_mag

    fx = golden("losses.npz")
    y, mask = torch.from_numpy(fx["y"]), torch.from_numpy(fx["mask"])
    sp = {"fft_sizes": [64, 128], "win_sizes": [64, 128], "hop_sizes": [16, 32], "logratio": 0}
    for causal in [False, True]:
        for cs in ([0] if not causal else [-8, -2, 0, 2, 8]):
            for lt in ["l1", "mse", "stft"]:
                x = torch.from_numpy(fx["x"]).requires_grad_(True)
                v = OracleFeatureLoss(lt, causal=causal, stft_params=sp)(x, y, mask=None if lt == "stft" else mask,
                                                                         causal_size=cs)
                v.backward()
                tag = f"{lt}_c{int(causal)}_cs{cs}"
                np.testing.assert_allclose(v.item(), float(fx[f"val_{tag}"]), rtol=1e-6, err_msg=tag)
                np.testing.assert_allclose(x.grad.numpy(), fx[f"grad_{tag}"], rtol=1e-5, atol=1e-9, err_msg=tag)
    # directly constructed STFTLoss(fft 32, win 20, hop 10, logratio .3): torch.stft sees them as named
    x = torch.from_numpy(fx["x"])
    win = torch.hann_window(20)
    xm, ym = stft_mag(x, 32, 10, 20, win), stft_mag(y, 32, 10, 20, win)
    v = 0.7 * (xm - ym).abs().mean() + 0.3 * (xm.log() - ym.log()).abs().mean()
    np.testing.assert_allclose(v.item(), float(fx["val_stftloss_direct"]), rtol=1e-6)
    v = multi_stft_loss(x, y, [32, 64], [32, 64], [8, 16], 0.25)
    np.testing.assert_allclose(v.item(), float(fx["val_ms_log"]), rtol=1e-6)


def test_oracle_stft_layer_and_scaler_match_reference():
    from oracle.modules import OracleLogMel

    fx = golden("stft_layer.npz")
    wav = torch.from_numpy(fx["wav"])[None]
    for center in [False, True]:
        o = OracleLogMel(fs=int(fx["fs"]), hop_size=128, fft_size=1024, win_length=1024, center=center)
        s = o.stft(wav)
        amp = torch.sqrt(s[..., 0] ** 2 + s[..., 1] ** 2).numpy()[:, ::8]
        np.testing.assert_allclose(amp, fx[f"amp_center{int(center)}"], rtol=1e-5, atol=1e-6)

    class Sc:
        mean_, var_ = fx["scaler_mean"], fx["scaler_var"]

    o = OracleLogMel(fs=int(fx["fs"]), hop_size=128, fft_size=1024, center=False, scaler=Sc)
    z = torch.from_numpy(fx["scaler_in"])
    np.testing.assert_allclose(((z - o.mean) / o.std).numpy(), fx["scaler_out"], rtol=1e-6, atol=1e-7)


def test_mel_basis_properties():
    """librosa is absent: the Slaney restatement is checked structurally (shape, band
    edges, area normalisation, non-negativity) and product == oracle copy."""
    from crank_amd.net.module.mlfb import slaney_mel_basis
    from oracle.modules import slaney_mel_basis as oracle_basis

    b = slaney_mel_basis(22050, 1024, 80, 80, 7600)
    assert b.shape == (80, 513) and b.dtype == np.float32 and (b >= 0).all()
    np.testing.assert_array_equal(b, oracle_basis(22050, 1024, 80, 80, 7600))
    freqs = np.linspace(0, 11025, 513)
    lo, hi = freqs[b[0] > 0], freqs[b[-1] > 0]
    assert lo.min() >= 80 - 22 and hi.max() <= 7600 + 1e-6
    # Slaney normalisation: each triangle integrates to ~1 over frequency
    area = (b * (freqs[1] - freqs[0])).sum(1)
    assert np.all(np.abs(area[10:] - 1) < 0.15)


def test_misc_goldens():
    from oracle.modules import _GRL, steplr_value

    fx = golden("misc.npz")
    x = torch.from_numpy(fx["grl_x"]).requires_grad_(True)
    y = _GRL.apply(x, 0.1)
    (y * torch.arange(8.0)).sum().backward()
    np.testing.assert_array_equal(y.detach().numpy(), fx["grl_y"])
    np.testing.assert_allclose(x.grad.numpy(), fx["grl_grad"], rtol=1e-7)
    for s, lr in zip(fx["steplr_steps"], fx["steplr_lr"]):
        assert abs(steplr_value(2e-4, int(s), 200000, 0.5) - float(lr)) < 1e-15


# ------------------------------------------------------------------ host logic without a GPU
def test_state_dict_layout_matches_reference_key_names_and_counts():
    """parameter counts 411008 / 450976 / 212352 / 212352 and the
    weight-norm key names (Appendix A.5), product layout == oracle modules."""
    from crank_amd.net.module.flat import net_keys, python_conv_table
    from oracle import pwg

    cases = [
        (0, dict(in_ch=80, out_ch=64, kernel_size=5, layers=8, stacks=4, aux_ch=0), 411008,
         pwg.ParallelWaveGANGenerator(80, 64, 5, 8, 4, aux_channels=0, upsample_conditional_features=False)),
        (0, dict(in_ch=128, out_ch=80, kernel_size=5, layers=8, stacks=4, aux_ch=34), 450976,
         pwg.ParallelWaveGANGenerator(128, 80, 5, 8, 4, aux_channels=34, upsample_conditional_features=False)),
        (0, dict(in_ch=64, out_ch=64, kernel_size=3, layers=6, stacks=3, aux_ch=0), 212352,
         pwg.ParallelWaveGANGenerator(64, 64, 3, 6, 3, aux_channels=0, upsample_conditional_features=False)),
        (1, dict(in_ch=113, out_ch=1, kernel_size=5, layers=8, stacks=4), None,
         pwg.ResidualParallelWaveGANDiscriminator(113, 1, 5, 8, 4)),
        (2, dict(in_ch=80, out_ch=14, kernel_size=5, layers=8), None, pwg.ParallelWaveGANDiscriminator(80, 14, 5, 8)),
        (2, dict(in_ch=128, out_ch=14, kernel_size=3, layers=3), None, pwg.ParallelWaveGANDiscriminator(128, 14, 3, 3)),
    ]
    for kind, kw, count, orac in cases:
        convs, total = python_conv_table(kind, **kw)
        mine = {k: shp for k, _, shp in net_keys(kind, convs)}
        ref = {k: tuple(v.shape) for k, v in orac.state_dict().items()}
        assert mine == ref, (set(mine) ^ set(ref))
        assert total == sum(int(np.prod(s)) for s in ref.values())
        if count is not None:
            assert total == count
        # offsets tile the flat block without gaps or overlap
        spans = sorted((off, off + int(np.prod(shp))) for _, off, shp in net_keys(kind, convs))
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_c_abi_exports_every_declared_symbol():
    lib_path = os.path.join(REPO, "crank_amd", "libcrank_hip.so")
    assert os.path.exists(lib_path), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    header = open(os.path.join(REPO, "include", "crank_hip.h")).read()
    declared = set(re.findall(r"\b(crk_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from crank_amd import _lib

    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib.crk_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.crk_version()


def _header_prototypes():
    """name -> (return type, [parameter types]) of every crk_* function include/crank_hip.h declares, as C type strings with
    the parameter names removed."""
    text = open(os.path.join(REPO, "include", "crank_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(crk_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"\b[A-Za-z_]\w*$", "", a).strip() if not a.endswith("*") else a  # drop the parameter's name
                params.append(re.sub(r"\s*\*", "*", " ".join(a.split())))
        protos[name] = (re.sub(r"\s*\*", "*", ret), params)
    return protos


def test_header_binding_and_library_agree_on_every_argument_list():
    """include/crank_hip.h, crank_amd/_lib.py SIGNATURES and the built library describe the same functions: for every
    declaration the C parameter list maps, position by position, onto the ctypes argtypes the binding installs (a pointer
    is a pointer, `int` is c_int, `long long` c_longlong, `unsigned long long` c_ulonglong, `float` / `double` by value
    c_float / c_double), and the return types agree.  A parameter added to one side only - what a name check cannot see -
    fails here, on CPU, before a mis-sized stack frame reaches the GPU."""
    from crank_amd import _lib

    protos = _header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    scalars = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong,
               "float": ctypes.c_float, "double": ctypes.c_double}

    def kind(ctype):  # "ptr" or the scalar ctypes class
        return "ptr" if ctype.endswith("*") else scalars[ctype.replace("const ", "")]

    def kind_of_ctypes(t):
        if t is ctypes.c_void_p or t is ctypes.c_char_p or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
            return "ptr"
        return t

    bad = []
    for name, (ret, params) in sorted(protos.items()):
        res, argtypes = _lib.SIGNATURES[name]
        want_ret = None if ret == "void" else kind(ret)
        got_ret = None if res is None else kind_of_ctypes(res)
        if want_ret != got_ret:
            bad.append((name, "return", ret, res))
        if len(params) != len(argtypes):
            bad.append((name, "arity", len(params), len(argtypes)))
            continue
        for i, (c, t) in enumerate(zip(params, argtypes)):
            if kind(c) != kind_of_ctypes(t):
                bad.append((name, i, c, t))
    assert not bad, bad
    assert len(protos) >= 60
    lib = ctypes.CDLL(os.path.join(REPO, "crank_amd", "libcrank_hip.so"))
    assert all(hasattr(lib, n) for n in protos)


def test_oracle_radam_follows_torch_radam_and_switches_regime_at_step_six():
    """oracle/optim.py::RAdam restates torch_optimizer.RAdam (crank/net/trainer/utils.py:44-45), a package absent here.
    The one independent statement of the same algorithm in this image is torch.optim.RAdam: same moments, same
    rectification term; it differs in where eps enters (after / before the bias correction of sqrt(v)) - invisible at
    |g| >> eps.  N_sma crosses 5 between steps 5 and 6 for beta2 = 0.999 (both implementations, whether they test
    >= 5 or > 5)."""
    from oracle.optim import RAdam

    torch.manual_seed(0)
    p0 = torch.randn(2000, dtype=torch.float64)
    a, b = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa, ob = RAdam([a], lr=2e-4), torch.optim.RAdam([b], lr=2e-4)
    regimes = []
    for i in range(12):
        g = torch.randn(2000, dtype=torch.float64) * (0.1 + i)
        a.grad, b.grad = g.clone(), g.clone()
        before = a.detach().clone()
        oa.step()
        ob.step()
        t = i + 1
        n_max = 2 / (1 - 0.999) - 1
        regimes.append(n_max - 2 * t * 0.999 ** t / (1 - 0.999 ** t) >= 5)
        np.testing.assert_allclose((a.detach() - before).numpy(), (b.detach() - before).numpy(), rtol=1e-6, atol=1e-14)
    assert regimes == [False] * 5 + [True] * 7
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=1e-9)


def test_oracle_lamb_first_step_by_hand():
    """oracle/optim.py::Lamb (pytorch_lamb.Lamb, crank/net/trainer/utils.py:46-47: absent package).  The first step from
    zero moments has a closed form: u = 0.1 g / (sqrt(0.001) |g| + 1e-6), w -= lr * min(||w||, 10) / ||u|| * u; an all-zero
    tensor moves by lr * u (ratio 1)."""
    from oracle.optim import Lamb, make_optimizer

    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(50, 40, dtype=torch.float64) * 2)  # norm ~ 89: clamped to 10
    z = torch.nn.Parameter(torch.zeros(30, dtype=torch.float64))
    opt = make_optimizer("lamb", [w, z], 1e-3)
    assert isinstance(opt, Lamb)
    w0 = w.detach().clone()
    w.grad, z.grad = torch.randn(50, 40, dtype=torch.float64), torch.randn(30, dtype=torch.float64)
    opt.step()
    u = 0.1 * w.grad / ((0.001 * w.grad ** 2).sqrt() + 1e-6)
    np.testing.assert_allclose(w.detach().numpy(), (w0 - 1e-3 * 10.0 / u.norm() * u).numpy(), rtol=1e-12)
    uz = 0.1 * z.grad / ((0.001 * z.grad ** 2).sqrt() + 1e-6)
    np.testing.assert_allclose(z.detach().numpy(), (-1e-3 * uz).numpy(), rtol=1e-12)
    assert opt.state[w]["trust_ratio"] == pytest.approx(10.0 / float(u.norm()))
    with pytest.raises(ValueError):
        make_optimizer("sgd", [w], 1e-3)


def test_product_refuses_to_run_without_gpu_or_library():
    import pytest

    from crank_amd.bin.train import get_model
    from crank_amd.utils import load_yaml

    with pytest.raises(RuntimeError):
        get_model(load_yaml(None), 2, device="cpu")


def test_config_surface_and_rejections():
    import pytest

    from crank_amd.net.trainer import TrainerWrapper
    from crank_amd.utils import load_yaml

    conf = load_yaml(None, trainer_type="lsgan", alpha={"l1": 3})
    assert conf["alpha"]["l1"] == 3 and conf["alpha"]["mse"] == 0 and conf["stft_params"]["fft_sizes"] == [64, 128]
    with pytest.raises(NotImplementedError):
        TrainerWrapper("nope")


def test_synthetic_batch_layout():
    from crank_amd.synthetic import make_batch

    b = make_batch(4, 50, 3, seed=1)
    assert b["in_feats"].shape == (4, 50, 80) and b["encoder_mask"].dtype == torch.bool
    for i in range(4):
        n = int(b["flen"][i])
        assert b["decoder_mask"][i, :n].all() and not b["decoder_mask"][i, n:].any()
        assert (b["org_h"][i, n:] == -100).all() and (b["org_h"][i, :n] >= 0).all()
        assert (b["in_feats"][i, n:] == 0).all()
        assert int(b["org_h"][i, 0]) != int(b["cv_h"][i, 0])


def test_oracle_conversion_equals_the_reference_conversion_golden():
    """tests/golden/convert_vqvae.npz - the REFERENCE's VQVAE2 converting 4 x 200 frames to the target speaker on given
    parameters (make_golden.py gen_convert) - against the oracle's VQVAE2 on the same parameters and inputs: decoded
    features and code indices (both sides are fp32 torch-CPU and run the same op sequence: bit for bit)."""
    import numpy as np
    import torch

    from crank_amd.utils import load_yaml
    from oracle.modules import OracleVQVAE2
    from tests.helpers import fill_models, golden, make_batch

    fx = golden("convert_vqvae.npz")
    B, T, S, seed = [int(v) for v in fx["meta_B_T_nspk_seed"]]
    conf = load_yaml(None)
    orac = OracleVQVAE2(conf, spkr_size=S).eval()
    fill_models({"G": orac})
    b = make_batch(B, T, S, in_dim=conf["input_size"], seed=seed)
    dec_h = torch.cat([b["cv_lcf0"], b["uv"]], -1)
    h = b["cv_h"].clone()
    h[:, :] = h[:, 0:1]
    with torch.no_grad():
        out = orac(b["in_feats"], None, dec_h, spkrvec=h, use_ema=False)
    assert np.abs(out["decoded"].numpy() - fx["decoded"]).max() <= 1e-6 * np.abs(fx["decoded"]).max()
    for i in range(2):
        assert np.array_equal(out["qidx"][i].numpy(), fx[f"qidx{i}"])
