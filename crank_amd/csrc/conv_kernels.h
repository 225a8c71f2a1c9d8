// Parameter blocks and launchers for the channel-last Conv1d kernels.
//
// Layout convention (MI355X-first, differs from the reference's (B,C,T) torch layout):
// every activation is "frames x channels" row-major fp32 in HBM, frame index
// n = b*T + t, explicit row stride (ld) so channel slices of wider buffers can be
// read/written in place.  A dilated Conv1d over time is an implicit GEMM
//   Y[n, co] = sum_tap sum_ci X[n + off0 + tap*dil, ci] * W[tap][co][ci]
// evaluated with v_mfma_f32_32x32x16_bf16 from LDS-staged bf16 tiles (fp32
// accumulate).  PRECISE mode splits both operands into bf16 hi+lo and issues three
// MFMAs per product (hi*hi + lo*hi + hi*lo), which restores ~fp32 accuracy on the
// same code path (used for parity; the fast path is the single hi*hi MFMA).
#pragma once
#include <vector>

#include "common.h"
#include "switches.h"

#define CRK_TM 128  // frames per workgroup tile (4 waves x 32 rows)

enum { MODE_PLAIN = 0, MODE_RESFWD = 1, MODE_BWDA = 2 };

struct ConvP {
  // input sources: channels [0,cinA) from xa, [cinA,cinA+cinB) from xb
  const float* xa; int lda; int cinA; float scaleA;
  const float* xb; int ldb; int cinB; float scaleB;
  int act_in; float slope;
  float drop_p; unsigned long long drop_seed;
  const unsigned long long* drop_seed_ptr;  // non-null: the call's seed lives in device memory, seeds above are ADDED to it
  int cin, cin_pad;
  // auxiliary (conditioning) source: one extra K=1 chunk (RESFWD only)
  const float* xc; int ldc; int cinC; int cinC_pad;
  // prepared weights (bf16 bit patterns)
  const uint16_t* w_hi; const uint16_t* w_lo;    // [ktaps][cout_pad][cin_pad]
  const uint16_t* wc_hi; const uint16_t* wc_lo;  // [cout_pad][cinC_pad]
  const uint16_t* w2_hi; const uint16_t* w2_lo;  // RESFWD: [128][64] rows 0-63 out, 64-127 skip
  const float* bias; const float* bias2a; const float* bias2b;
  int cout, cout_pad, ktaps, dil, off0;
  int B, T, tiles_per_utt;
  // PLAIN epilogue
  float* y; int ldy; int accumulate; int act_out; float out_scale;
  const float* dmask; int ldm; int dmask_act;
  const float* res; int ldr; float res_scale;
  float epi_drop_p; unsigned long long epi_drop_seed;  // multiply by the regenerated dropout keep-scale
  // RESFWD epilogue
  float* skip; int skip_init; float* sv_ta; float* sv_sb; float* sv_z;
  // BWDA epilogue
  const float* ta; const float* sb;
  // LDS carve-up (bytes)
  int xs_stride, cs_stride, ws_stride, zs_stride;
  int o_xlo, o_chi, o_clo, o_whi, o_wlo, o_zhi, o_zlo;
  int lds_bytes;
  int dbg;  // ablation switches for kernel timing experiments (CRK_DBG env, 0 in production)
};

struct WgradP {
  // A operand: dY, channels [0,ca1) from a1, [ca1, ca1+ca2) from a2
  const float* a1; int lda1; int ca1; float sa1;
  const float* a2; int lda2; int ca2; float sa2;
  int ca, ca_pad;
  // B operand: conv input (shifted per tap) with the forward prologue
  const float* x; int ldx; int cx; int cx_pad; float sx; int act_in; float slope;
  float drop_p; unsigned long long drop_seed; const unsigned long long* drop_seed_ptr;
  // optional aux operand handled as tap index == ktaps
  const float* xc; int ldc; int cc; int cc_pad; int has_aux;
  int ktaps, dil, off0;
  int B, T;
  float* partial;       // [B][ktaps][ca][cx]
  float* partial_aux;   // [B][ca][cc]
  float* bias_partial;  // [B][ca]
  // table-entry view: which taps of the problem this entry covers (or its aux 1x1)
  int grp_tap0, grp_ntap, grp_aux;
  int cpg, ngroups;  // 64-frame chunks per partial-sum group, number of groups (workgroups beyond it exit)
  int dbg;
};

// one weight-normalised Conv1d of a network (device-visible table entry)
struct ConvEntry {
  int cout, cin, k;
  long long off_g, off_v, off_b;  // element offsets into the net's flat fp32 parameter block (off_b<0: no bias)
  // forward-layout prepared weights [k][fw_rows][fw_kp] (element offset into wprep)
  long long fw_off; int fw_rows, fw_kp, fw_row0;
  // data-gradient layout [k][bw_rows][bw_kp], tap-flipped + transposed
  long long bw_off; int bw_rows, bw_kp, bw_col0;
  // weight-gradient partial sums
  long long pt_off; int pt_rows, pt_row0, pt_cx, pt_taps, pt_tap0; float pt_scale;
  int pt_groups;  // partial-sum slots of this entry (device table); host table: 1 = lives in the stack region
  long long pb_off;  // bias partial [G][pt_rows]
  long long norm_off;
  // MFMA-fragment-ordered copy for the channel-split stack kernels (stack2_kernels.hip): 1 KB per (tap, 32-row
  // tile, 16-wide k step), lane l's 8 bf16 = A[row l&31][k 8*(l>>5)..+8] at 16*l.  fr_mode: 0 none, 1 gated conv
  // (tile mt = tanh rows 16mt.. | sigmoid rows 64+16mt..), 2 its conditioning 1x1 (same rows, K padded to 64),
  // 3 out 1x1 (tiles 0,1), 4 skip 1x1 (tiles 2,3 of the same [4][4] block), 5 plain conv: [tap][32-row tile][kp/16][64][8],
  // 6 plain conv of a kind-2 chain (pstack2_kernels.hip): [32-row tile][tap][kp/16][64][8] - a tile's fragments are one run
  long long fr_off; int fr_mode;
  // ... and of the data-gradient (transposed, tap-flipped) layout for the channel-split backward (stack2b_kernels.hip):
  // A[row = input channel][k = bw_col0 + output channel] as [tap][bw_rows / 32][bw_kp / 16][64 lanes][8]; < 0: none.
  // bfr_mode 1 (kind-2 chains): [bw_rows / 32][tap][bw_kp / 16][64][8]
  long long bfr_off; int bfr_mode;
};

// ---- fused multi-layer forward of the gated residual blocks (stack_kernels.hip) ----
struct StackLayer {
  long long f_conv, f_aux, f_os;   // fragment-ordered copies (stack2_kernels.hip): [k][4][4][64][8], [4][4][64][8], [4][4][64][8]
  long long w_conv, w_aux, w_os;   // element offsets of the bf16 planes: [k][128][64], [128][aux_pad], [128][64]
  long long b_conv, b_out, b_skip; // parameter offsets of the biases (-1: none)
  int dil, off0;
};
struct StackP {
  const float* x0;     // block-0 input [N,64] (first conv output)
  const float* c; int ldc; int aux_ch, aux_pad;
  float* saved;        // X | TA | SB | Z planes (null: nothing is saved); the fused kernel writes TA and SB
  uint16_t *xb_hi, *xb_lo;  // [L][N,64] bf16 block inputs as the conv sees them (weight-gradient operand)
  uint16_t *zb_hi, *zb_lo;  // [L][N,64] bf16 gate outputs
  uint16_t *tb_hi, *tb_lo, *sg_hi, *sg_lo;  // [L][N,64] bf16 tanh / sigmoid halves of the gate (gate backward)
  uint16_t *cb_hi, *cb_lo;  // [N,aux_pad] bf16 conditioning
  float* skip;         // [N,64] running skip sum (output)
  const float* params; // the net's flat parameter block (biases)
  const uint16_t* whi; const uint16_t* wlo;
  const StackLayer* layers;  // device table [L]
  int B, T, L, ktaps;
  int hl, hr, max_off;  // receptive-field halo of the whole stack (frames), largest tap offset
  int tmo, tiles_per_utt;
  float drop_p; unsigned long long drop_seed; const unsigned long long* drop_seed_ptr;
  int o_xlo, o_zhi, o_zlo, o_chi, o_clo, o_whi, o_wlo, o_bias, o_tab, w_bytes, lds_bytes;
  int nw;  // waves per workgroup (window = 32*nw frames)
  int dbg; // ablation switches (experiments only; 0 in production)
  int ft, fh;  // stack2: 32-frame tiles per wave, frame halves per workgroup (window = 32*ft*fh frames, 4*fh waves)
  int ft1;     // stack2, fh = 2: tiles per wave of frame half 1 when it differs from ft (window = 32*(ft+ft1) frames); 0: ft
  int o_zs, o_cs;  // stack2 LDS carve-up: gate-output tile, conditioning tile
  int o_xf;        // stack2, folded first conv: the stack-input tile [rows][kp_first] bf16 (row stride kp_first * 2 + 16)
  // stack2 with the first conv and the head folded in (generator stacks): x_in != null selects it
  const float* x_in; int ldx_in, in_ch, kp_first;   // stack input [N, in_ch] fp32; first-conv reduction width (in_ch padded to 16)
  long long f_first, b_first;                      // fragment-ordered first-conv weights [2][kp_first/16][64][8]; bias offset in params
  long long f_h1, b_h1, f_h2, b_h2;                // head: 64 -> 64 ([2][4][64][8]) and 64 -> out_ch ([tiles][4][64][8])
  float* y; int ldy, out_ch; float head_scale;     // stack output [N, out_ch] fp32; sqrt(1 / L)
  uint16_t* fin_hi; uint16_t* head_hi;             // saved bf16 planes: first-conv input [N, kp_first]; head operands S | H1 ([N,64] each)
  // tanh / sigmoid planes in the layout of the channel-split data-gradient chain (stack2b_kernels.hip): ts_stride > 0 selects it,
  // plane l starts at tb_hi / sg_hi + l * ts_stride.  Blocks of 32 consecutive frames (n >> 5) x 2 channel groups (z channels
  // 32 mt2 ..) x 2 pieces x 64 lanes x 16 bytes: lane = (half, n & 31), piece g of group mt2 = the accumulator-layout quads
  // 2g, 2g + 1 (channels 32 mt2 + 16 g + 8 q' + 4 half + j) - what a lane of either kernel holds in registers, so producer and
  // consumer move whole 1 KB runs (full cache lines) without a lane exchange.
  long long ts_stride;
  int cs_stride;  // stack2x (split-operand forward): row stride in bytes of the packed conditioning tile (one dword per channel)
};
// ---- fused chains of plain convs, either direction (pstack_kernels.hip) ----
struct PsLayer {
  long long w_off;      // bf16 operand-plane offset of this conv: [k][rows_pad][kp]
  long long b_off;      // bias offset in the parameter block (-1: none)
  int rows, rows_pad;   // output channels (valid / padded to 32)
  int kp;               // reduction width: input channels padded to 16
  int k, dil, off0;     // taps, dilation, frame offset of tap 0
  int epi;              // on the layer output: 0 none, 1 ReLU, 2 LeakyReLU, 3 x ReLU'(plane), 4 x LeakyReLU'(plane)
  int mask_w;           // (epi 3/4) row width of the plane whose sign selects the derivative
  long long mask_plane; // (epi 3/4) its element offset from PsP::mask_hi
  long long save_plane; // element offset (from PsP::save_hi / save_lo) of the plane receiving this layer's INPUT operand
  long long f_off;      // the weights in fragment order, [tile][tap][kp / 16][64 lanes][8] (pstack2_kernels.hip); < 0: none
};
struct PsP {
  const float* x; int ldx, cin; float in_scale; int in_act;  // layer-0 operand: act(in_scale * x), fp32 rows
  float* y; int ldy; float out_scale;                        // chain output [N, rows of the last layer] fp32
  const float* params; const uint16_t *whi, *wlo;
  uint16_t *save_hi, *save_lo;   // operand planes (null: not kept)
  const uint16_t* mask_hi;       // planes the mask epilogues read
  const PsLayer* layers; int L;  // device table; `tail`: entry L describes only the plane that receives layer L-1's
  int tail;                      // (masked) output - the chain ends in a saved operand instead of an fp32 output
  int B, T; float slope;
  int hl, hr, tmo, tiles_per_utt, nw, os;
  int o_olo, o_whi, o_wlo, o_bias, o_tab, w_bytes, lds_bytes;
  double algo_bytes;  // algorithmic HBM bytes of the launch (pstack_plan; measurement only)
  int os_b;           // pstack2: row stride of the second operand tile (at LDS offset o_olo)
  const float* in_num; const float* in_den;  // (both or none) in_scale is multiplied by in_num[0] / in_den[1], read on the device
  // pstack2: layer 0's table entry as kernel arguments (pstack2_plan copies it) - the first weight fragments and the operand
  // rows are requested by the kernel's first instructions instead of behind a load of the device table
  long long l0_f_off, l0_save_plane; int l0_k, l0_kp, l0_rows_pad;
};
struct PwLayer {  // weight gradient of one plain conv on bf16 planes
  long long a_hi, a_lo;         // output-gradient plane [N, wa]: element offsets from PwP::abase
  long long b_hi, b_lo;         // input-operand plane [N, wb]: element offsets from PwP::bbase
  int wa, wb;                   // plane row widths (multiples of 16)
  int ca, cb;                   // valid channels: cout, cin
  int k, dil, off0;
  long long pt, pb;             // float offsets of the partial sums / bias sums (pb < 0: no bias)
};
#define CRK_MAX_NETS_PW 8
struct PwP {
  const PwLayer* layers;  // device table
  const uint16_t* abase;  // bf16 planes of the backward chain (output gradients)
  const uint16_t* bbase;  // bf16 planes of the forward chain (input operands)
  float* partials;
  int B, T, cpg, G;       // 64-frame chunks per group, number of groups
};
struct PwMP { PwP q[CRK_MAX_NETS_PW]; int first[CRK_MAX_NETS_PW + 1]; int n; };  // several nets' tables in one launch
int launch_pstack_wgrad_multi(const PwMP& m, int total_layers, int max_G, int max_wa, int max_wb, int max_tiles, double flops,
                              double bytes, hipStream_t s);  // max_tiles: most (tap, cin band, cout band) tiles of any conv
int pstack_wgrad_supported(int ca, int cb, int wa, int wb, int k, int dil);
int launch_pstack_wgrad(const PwP& p, int nlayers, int max_wa, int max_wb, bool precise, double flops, hipStream_t s, int max_tiles = 0);
int pstack_plan(PsP& p, const PsLayer* host_layers, bool precise);
int launch_pstack(const PsP& p, bool precise, double flops, hipStream_t s);
int pstack2_plan(PsP& p, const PsLayer* host_layers);  // channel-split chains (plain bf16); CRK_ERR_UNSUPPORTED: use pstack
int launch_pstack2(const PsP& p, double flops, hipStream_t s);
// ... forward chains in split-operand (bf16x3) arithmetic (pstack2x_kernels.hip): the forward of the bf16x3f mode; hi planes only
int pstack2x_plan(PsP& p, const PsLayer* host_layers);
int launch_pstack2x(const PsP& p, double flops, hipStream_t s);

// ---- fused data-gradient chain of the gated residual blocks (stack_kernels.hip) ----
struct StackBLayer {
  long long w_os, w_conv, w_aux;  // element offsets of the data-gradient planes: [64][128], [k][64][128], [64][128]
  int dil, off0;                  // frame offset of tap 0 of the transposed conv
  long long f_os, f_conv, f_aux;  // the same weights in MFMA-fragment order (ConvEntry::bfr_off): [2][8][64][8], [k][2][8][64][8]
};
struct StackBP {
  const float* dS;       // [N,64] gradient wrt the skip sum
  const float* saved;    // fp32 planes of the forward (X_0: LeakyReLU mask of the discriminator)
  const uint16_t *tb_hi, *tb_lo, *sg_hi, *sg_lo;  // [L][N,64] bf16 tanh / sigmoid planes of the forward
  float* dX0;            // [N,64] fp32 dX_0 (gradient wrt the stack input)
  uint16_t *gb_hi, *gb_lo;    // [L][N,128] bf16 gate pre-activation gradients dG_l
  uint16_t *dxb_hi, *dxb_lo;  // [L][N,64] bf16 dX_l (plane l+1 is the out-conv gradient operand of block l)
  uint16_t *dsb_hi, *dsb_lo;  // [N,64] bf16 dS
  float* dc; int lddc; int aux_ch;  // conditioning gradient (null: not wanted)
  const uint16_t* whi; const uint16_t* wlo;
  const StackBLayer* layers;  // device table [L]
  int B, T, L, ktaps;
  int hl, hr, max_off;
  int tmo, tiles_per_utt;
  float drop_p; unsigned long long drop_seed; const unsigned long long* drop_seed_ptr;
  int mask_l0; float slope;   // discriminator: dX_0 *= LeakyReLU'(X_0)
  int o_glo, o_whi, o_wlo, w_bytes, lds_bytes, nw;
  // generator stacks, plain bf16: the head's data gradient (dy -> dS) in front of the chain and the first conv's
  // (dX_0 -> dx) behind it, in the same launch (dy != null selects it)
  const float* dy; int lddy, out_ch, kp_y;          // gradient wrt the stack output [N, out_ch]; out_ch padded to 16
  long long w_h2, w_h1, w_first;                    // data-gradient planes: [64][kp_y], [64][64], [in_rows][64]
  const uint16_t* hmask_hi;                         // forward head planes S | H1 ([N,64] each): ReLU masks
  uint16_t* hb_hi;                                  // head gradient planes: bf16 dy [N, kp_y], then G1 [N,64]
  float head_scale;                                 // sqrt(1 / L)
  float* dx; int lddx, in_ch, in_rows; float dx_scale;  // gradient wrt the stack input (null: not wanted); in_ch padded to 32
  // channel-split kernel (stack2b_kernels.hip): fragment-ordered head / first-conv weights, window shape, LDS carve-up
  long long f_h2, f_h1, f_first;
  int ft, o_dx, o_tab;
  long long ts_stride;  // tanh / sigmoid planes: element stride between blocks (the lane-record layout, StackP::ts_stride)
  // 1: the planes this chain WRITES (dG_l, dX_l, dS) in 4-frame records - the 16-byte piece of frame n, channels 8c .. 8c+7
  // at byte (n >> 2) * 8 W + c * 64 + (n & 3) * 16 of a plane of W channels.  Row-major, a wave's store (32 frames, 16 bytes
  // each from two half-waves) is 32 bytes in each of 32 rows: 32 partial-line requests; as records it is 8 whole 128-byte
  // lines (4 frames x the two halves' adjacent pieces), and the weight gradient (stack_wgrad_kernel, StackWP::rec_g) still
  // reads 1 KB of consecutive bytes per wave and writes its LDS tile without bank conflicts.  (32-frame records - 1 KB per
  // store - were measured first: the chain as fast, the weight gradient + 9 %: its lanes then either read pieces 512 bytes
  // apart or write 8 rows of one piece to LDS, a two-way bank conflict.)  dX_0 stays row-major: the first conv's weight
  // gradient (the plain convs' kernel) reads it.  Needs B * T % 4 == 0.  The same for the forward's block-input and
  // gate-output planes was measured and dropped: no change (two waves per SIMD hide the forward's stores).
  int rec;
  int dbg;  // timing experiments only (CRK_S2B_DBG; bit 0: plane stores dropped by the bounds check)
};
// ---- weight gradients of the gated residual blocks from the bf16 planes (stack_kernels.hip) ----
struct StackWLayer {
  long long pt_conv, pb_conv, pt_os, pb_os, pt_aux;  // float offsets into the partial-sum block (pb < 0: no bias)
  int dil, off0;
};
struct StackWP {
  const uint16_t *xb_hi, *xb_lo, *zb_hi, *zb_lo, *cb_hi, *cb_lo;   // forward planes (cb null: no conditioning)
  const uint16_t *gb_hi, *gb_lo, *dxb_hi, *dxb_lo, *dsb_hi, *dsb_lo;  // backward planes
  const StackWLayer* layers;  // device table [L]
  float* partials;
  int B, T, L, ktaps, aux_ch, aux_pad, cpg, G;  // cpg: 64-frame chunks per group (runs over the utterances)
  int rec_g;  // the backward planes (gb, dxb[1..], dsb) are 4-frame records (StackBP::rec)
};
int stack_wgrad_supported(int ktaps, int max_dil, int aux_ch);
int launch_stack_wgrad(const StackWP& p, bool precise, hipStream_t s);
int stack_bwd_plan(StackBP& p, bool precise);
int stack2_bwd_plan(StackBP& p);
int launch_stack2_bwd(const StackBP& p, hipStream_t s);
int stack_bwd_waves(bool precise);
int launch_stack_bwd(const StackBP& p, bool precise, hipStream_t s);
int stack_fwd_plan(StackP& p, bool precise);
int launch_stack_fwd(const StackP& p, bool precise, hipStream_t s);
// channel-split variant (stack2_kernels.hip; plain bf16 only): plan fills ft / tmo / tiles_per_utt / lds_bytes
int stack2_fwd_plan(StackP& p);
int launch_stack2_fwd(const StackP& p, hipStream_t s);
// ... in split-operand (bf16x3) arithmetic, generator stacks only (stack2x_kernels.hip): the forward of the bf16x3f mode; same
// windows and the same saved hi planes as stack2_fwd_kernel, so the plain-bf16 backward kernels follow it unchanged
int stack2x_fwd_plan(StackP& p);
int launch_stack2x_fwd(const StackP& p, hipStream_t s);
#define CRK_PROF_CLASSES 9  // 0-6: conv kernels (crank_hip.h); 7 the VQ search; 8 the on-the-fly log-mel kernel
void conv_prof_begin(int cls, double flops, hipStream_t s);
void conv_prof_end(int cls, hipStream_t s);
void conv_prof_bytes(int cls, double bytes);

void conv_fill_lds(ConvP& p, int mode, bool precise);
int launch_conv(const ConvP& p, int mode, bool precise, hipStream_t s);
int wgrad_expand(const WgradP& job, bool precise, std::vector<WgradP>& out);
int launch_wgrad_table(const WgradP* d_jobs, const std::vector<WgradP>& h_jobs, int B, int T, int max_groups, bool precise,
                       hipStream_t s);
// several nets in one launch of the weight preparation / the weight-norm backward (crk_nets_prepare, crk_nets_wnorm_bwd)
#define CRK_MAX_NETS 8
struct NetRef {
  const ConvEntry* ents; const float* params; float* grads; const float* partials; float* norms;
  uint16_t *whi, *wlo;
  int n_ents, first;  // entries of this net; index of its first workgroup column in the launch
};
struct NetRefs { NetRef r[CRK_MAX_NETS]; int n; float* bump; };  // bump: an Adam step count advanced by the prep launch
int launch_weight_prep_multi(const NetRefs& R, int total_entries, int nmax, hipStream_t s);  // nmax: largest cin * k
int launch_step_bump(float* step, hipStream_t s);
int launch_wnorm_bwd_multi(const NetRefs& R, int total_entries, hipStream_t s);
int launch_weight_prep(const ConvEntry* d_entries, int n_entries, int nmax, const float* params, uint16_t* wprep_hi,
                       uint16_t* wprep_lo, float* norms, hipStream_t s);
int launch_wnorm_bwd(const ConvEntry* d_entries, int n_entries, const float* params, float* grads,
                     const float* partials, const float* norms, hipStream_t s);
int conv_kernels_init();
