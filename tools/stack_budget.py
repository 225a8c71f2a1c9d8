"""Byte / FLOP budget of the four generator stacks (enc0, enc1, dec1, dec0; crank/net/module/vqvae2.py:237-273) forward +
data gradient + weight gradient at the benchmark shape (B = 64 x T = 500 = 32 000 frames): what `stacks_alone` in bench.py
times.  Pure arithmetic, no GPU: prints the markdown tables of DESIGN.md section 4 ("The budget").

    python tools/stack_budget.py

Planes are bf16 [N, C] (2 B per element).  Per gated block and frame the CURRENT scheme moves
  forward        W: Xb 128 (block input as the conv saw it), tanh 128, sigmoid 128, z 128
  data gradient  R: tanh 128, sigmoid 128 (x window rows / own rows: the halo is re-read)   W: dG 256, dX 128
  weight grad.   R: dG 256, Xb 128, z 128, dX 128 (dS 128: one plane per stack, re-read by every block out of L2 / the
                    infinity cache - counted once)                                        W: fp32 partial sums per group
  weight norm    R: the partial sums
Peaks: 2 500 TFLOP/s dense bf16 MFMA, HBM 8 TB/s spec / 6.3 TB/s achievable (MI355X_MICROARCH.md; a float4 copy measures 6.29).
"""
N = 32000
T = 500
B = 64
PEAK = 2500e12
HBM = 6.3e12

# name, in_ch, out_ch, k, blocks, aux, dilations
STACKS = [("enc0", 80, 64, 5, 8, 0), ("enc1", 64, 64, 3, 6, 0), ("dec1", 64, 64, 3, 6, 0), ("dec0", 128, 80, 5, 8, 34)]
GROUPS = 32  # utterance groups of the weight-gradient kernel (partial sums per group)


def halo(k, L):
    # layers = 2 per dilation cycle (1, 2), `stacks` cycles: (k - 1) / 2 * (1 + 2) * L / 2 frames per side
    return (k - 1) // 2 * 3 * (L // 2)


def rows(k, direction):
    # window rows of the shipped kernels: 192 (k = 5 both directions; k = 3 data gradient), 160 (k = 3 forward)
    return 160 if (k == 3 and direction == "fwd") else 192


def stack_numbers(name, cin, cout, k, L, aux):
    mac_block = k * 64 * 128 + aux * 128 + 128 * 64
    mac_ends = cin * 64 + 64 * 64 + 64 * cout
    mac = L * mac_block + mac_ends
    h = halo(k, L)
    tmo = 125  # 4 windows per utterance on 256 workgroups
    out = {"name": name, "mac": mac, "halo": h}
    # MFMA work actually issued: every row of the window, padded channels (aux 34 -> 48, out 80 -> 96, in 80 -> 80)
    pad = lambda c, m: (c + m - 1) // m * m  # noqa: E731
    mac_block_issued = k * 64 * 128 + pad(aux, 16) * 128 + 128 * 64
    mac_ends_issued = pad(cin, 16) * 64 + 64 * 64 + 64 * pad(cout, 32)
    for d in ("fwd", "dgrad"):
        R = rows(k, d)
        out["issued_" + d] = (L * mac_block_issued + mac_ends_issued) * R / tmo
        out["rows_" + d] = R
        out["own_" + d] = tmo / R
        out["need_" + d] = (tmo + 2 * h) / R
    out["issued_wgrad"] = L * mac_block_issued + mac_ends_issued  # no halo: the reduction runs over the frames themselves
    # ---- bytes per frame of the CURRENT scheme ----
    ends_fwd = 4 * cin + 4 * cout + 2 * pad(cin, 16) + 2 * 128 + (4 * aux + 2 * pad(aux, 16) if aux else 0)  # x, y, first-conv plane, head planes, c
    ends_dg = 4 * cout + 4 * cin + 2 * 128 + 128 + 2 * pad(cout, 16) + 2 * 128 + (4 * aux if aux else 0)   # dy, dx, head masks R, dS W, head grads W
    ends_wg = 2 * pad(cin, 16) + 128 + 2 * 128 + 2 * 128 + 2 * pad(cout, 16)                                # first conv + head operands
    reread = (tmo + 2 * h) / tmo  # the data gradient loads the gate planes of every window row inside the utterance
    par_block = (k * 64 * 128 + aux * 128 + 128 * 64) * 4 * GROUPS / N  # fp32 partial sums per frame
    cur = {"fwd": L * 512 + ends_fwd,
           "dgrad": L * (256 * reread + 384) + ends_dg,
           "wgrad": L * (640 + par_block) + 128 + ends_wg,
           "wnorm": L * par_block}
    out["cur"] = cur
    # ---- candidates ----
    # (a) gates recomputed in the data gradient from Xb (+ conditioning): forward stops writing tanh / sigmoid, the chain reads
    #     Xb with its halo instead of both gate planes; + the dilated conv (+ conditioning) once more on the matrix pipe
    a = dict(cur)
    a["fwd"] = L * 256 + ends_fwd
    a["dgrad"] = L * (128 * reread + 384) + ends_dg + (2 * pad(aux, 16) * reread if aux else 0)
    out["a"] = a
    out["a_extra_mac"] = L * (k * 64 * 128 + pad(aux, 16) * 128) * rows(k, "dgrad") / tmo
    # (b) weight gradients of the out|skip 1x1 pair (and of the conditioning conv) inside the data-gradient launch: the chain
    #     stops writing dX, the weight-gradient kernel stops reading z / dX / dS; every WORKGROUP (256 of them, 125 frames each)
    #     writes its own partial sums of those convs instead
    bb = dict(cur)
    par_b_wg = (128 * 64 + aux * 128) * 4 * 256 / N  # per frame: one partial set per workgroup and block
    bb["dgrad"] = L * (256 * reread + 256 + par_b_wg) + ends_dg
    bb["wgrad"] = L * (384 + k * 64 * 128 * 4 * GROUPS / N) + ends_wg
    bb["wnorm"] = L * (k * 64 * 128 * 4 * GROUPS / N + par_b_wg)
    bb["fwd"] = L * 384 + ends_fwd  # z is no longer read by anyone (the chain forms tanh * sigmoid itself)
    out["b"] = bb
    # (c) the gate planes of the halo rows exchanged on chip instead of re-read (not possible between workgroups: listed as the bound)
    c = dict(cur)
    c["dgrad"] = L * (256 + 384) + ends_dg
    out["c"] = c
    # all three
    f = dict(cur)
    f["fwd"] = L * 128 + ends_fwd
    f["dgrad"] = L * (128 + 256 + par_b_wg) + ends_dg
    f["wgrad"] = bb["wgrad"]
    f["wnorm"] = bb["wnorm"]
    out["abc"] = f
    return out


def main():
    S = [stack_numbers(*s) for s in STACKS]
    alg = 2.0 * N * sum(s["mac"] for s in S)
    print(f"algorithmic FLOP of one pass: 3 x {alg / 1e9:.1f} GFLOP = {3 * alg / 1e9:.1f} GFLOP; 30 % of the dense bf16 peak = "
          f"{3 * alg / (0.3 * PEAK) * 1e6:.0f} us, 20 % = {3 * alg / (0.2 * PEAK) * 1e6:.0f} us, 16 % = {3 * alg / (0.16 * PEAK) * 1e6:.0f} us\n")
    print("| stack | blocks, k, halo/side | forward MAC / frame | window rows fwd / dgrad (own frames 125) | MFMA time at peak incl. window rows and padded channels: fwd / dgrad / wgrad (us) |")
    print("|---|---|---|---|---|")
    tot_issue = 0.0
    for s, st in zip(S, STACKS):
        tf, td, tw = (2.0 * N * s["issued_" + d] / PEAK * 1e6 for d in ("fwd", "dgrad", "wgrad"))
        tot_issue += tf + td + tw
        print(f"| {s['name']} | {st[4]}, {st[3]}, {s['halo']} | {s['mac']:,} | {s['rows_fwd']} / {s['rows_dgrad']} | {tf:.1f} / {td:.1f} / {tw:.1f} |")
    print(f"\nMFMA time of the pass at peak, as issued: {tot_issue:.0f} us (algorithmic: {3 * alg / PEAK * 1e6:.0f} us) -> "
          f"{tot_issue / (3 * alg / PEAK * 1e6):.2f}x the algorithmic work is on the matrix pipe\n")

    def table(key, title):
        print(f"**{title}**\n")
        print("| stack | forward MB | data gradient MB | weight gradient MB | weight norm MB | sum MB | us at 6.3 TB/s |")
        print("|---|---|---|---|---|---|---|")
        tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "wnorm": 0.0}
        for s in S:
            d = s[key]
            mb = {k: v * N / 1e6 for k, v in d.items()}
            for k in tot:
                tot[k] += mb[k]
            sm = sum(mb.values())
            print(f"| {s['name']} | {mb['fwd']:.0f} | {mb['dgrad']:.0f} | {mb['wgrad']:.0f} | {mb['wnorm']:.0f} | {sm:.0f} | {sm * 1e6 / HBM * 1e6:.0f} |")
        sm = sum(tot.values())
        t_hbm = sm * 1e6 / HBM * 1e6
        print(f"| **all four** | **{tot['fwd']:.0f}** | **{tot['dgrad']:.0f}** | **{tot['wgrad']:.0f}** | **{tot['wnorm']:.0f}** | **{sm:.0f}** | **{t_hbm:.0f}** |")
        return sm, t_hbm

    res = {}
    res["cur"] = table("cur", "Current scheme (designed bytes; measured by PMC: see the text)")
    print()
    res["a"] = table("a", "(a) gates recomputed in the data-gradient chain")
    extra = 2.0 * N * sum(s["a_extra_mac"] for s in S) / PEAK * 1e6
    print(f"\n(a) puts {extra:.0f} us more MFMA time at peak on the chain ({sum(s['a_extra_mac'] for s in S) / sum(s['issued_dgrad'] for s in S) * 100:.0f} % of its issued work)\n")
    res["b"] = table("b", "(b) out|skip (and conditioning) weight gradients inside the data-gradient launch")
    print()
    res["c"] = table("c", "(c) gate planes without the halo re-read")
    print()
    res["abc"] = table("abc", "(a) + (b) + (c) together: the floor of the plane scheme")
    print("\n| scheme | GB per pass | HBM time at 6.3 TB/s | MFMA time at peak as issued | ceiling if the two overlap perfectly | ceiling if they add |")
    print("|---|---|---|---|---|---|")
    t_alg = 3 * alg / PEAK * 1e6
    for key, title in (("cur", "current"), ("a", "(a)"), ("b", "(b)"), ("c", "(c)"), ("abc", "(a)+(b)+(c)")):
        sm, th = res[key]
        tm = tot_issue + (extra if key in ("a", "abc") else 0.0)
        print(f"| {title} | {sm / 1e3:.2f} | {th:.0f} us | {tm:.0f} us | {t_alg / max(th, tm) * 100:.0f} % | {t_alg / (th + tm) * 100:.0f} % |")


if __name__ == "__main__":
    main()
