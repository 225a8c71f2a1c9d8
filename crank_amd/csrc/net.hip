// Host-side orchestration of the convolutional networks of the step, exposed through
// the C ABI declared in include/crank_hip.h.
//
// A "net" is one of the three third-party stacks crank instantiates (SURVEY.md
// Appendix A; call sites crank/net/module/vqvae2.py:237-273, spkradv.py:49-60,
// crank/bin/train.py:78-128):
//   kind 0  gated-residual generator  (ParallelWaveGANGenerator, ReLU head, optional aux)
//   kind 1  gated-residual discriminator (ResidualParallelWaveGANDiscriminator, LeakyReLU)
//   kind 2  plain dilated conv stack + LeakyReLU (ParallelWaveGANDiscriminator)
// All parameters of a net live in one flat fp32 block (weight_g / weight_v / bias per
// conv); the handle owns the weight-normalised bf16 operand planes, the per-utterance
// weight-gradient partials and the backward scratch.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "conv_kernels.h"

struct crk_net_desc {
  int kind;
  int in_ch, out_ch, kernel_size, layers, stacks;
  int res_ch, gate_ch, skip_ch, aux_ch;
  int conv_ch;
  int causal;
  int use_bias;
  float slope;
  float dropout;
};

enum { ROLE_FIRST = 0, ROLE_CONV = 1, ROLE_AUX = 2, ROLE_OUT = 3, ROLE_SKIP = 4, ROLE_LAST1 = 5, ROLE_LAST2 = 6, ROLE_PLAIN = 7 };

struct ConvMeta {
  int role, layer, dilation;
};

// The compute entry points (crk_net_forward / _backward*) never allocate: every device buffer and table a batch shape needs is
// made by crk_net_reserve(net, B, T), outside the step ("takes raw device pointers, sizes and a stream;
// never allocates").  g_may_alloc is true only inside crk_net_create / crk_net_reserve; an allocation site reached with it
// false returns CRK_ERR_ARG (the shape was not reserved).  g_net_allocs counts the allocations (crk_debug_alloc_count).
static bool g_may_alloc = false;
static long long g_net_allocs = 0;
static hipError_t net_malloc_(void** p, size_t bytes) {
  if (!g_may_alloc) return hipErrorInvalidValue;
  g_net_allocs++;
  return hipMalloc(p, bytes);
}
#define NET_MALLOC(pp, bytes) net_malloc_(reinterpret_cast<void**>(pp), (bytes))
struct AllocScope {
  bool was;
  AllocScope() : was(g_may_alloc) { g_may_alloc = true; }
  ~AllocScope() { g_may_alloc = was; }
};
static int not_reserved(const char* what) {
  fprintf(stderr, "[crank_hip] %s: this batch shape needs device buffers / tables the handle does not hold - call "
                  "crk_net_reserve(net, B, T) once per batch shape, outside the step (the compute entry points never allocate)\n", what);
  return CRK_ERR_ARG;
}

struct Net {
  crk_net_desc d;
  std::vector<ConvEntry> ents;
  std::vector<ConvMeta> meta;
  ConvEntry* d_ents = nullptr;
  long long n_params = 0;
  long long wprep_elems = 0, norm_elems = 0;
  uint16_t *whi = nullptr, *wlo = nullptr;
  float* norms = nullptr;
  unsigned long long prepared_version = ~0ull;
  const float* prepared_params = nullptr;
  // weight-norm backward deferred by CRK_FLAG_DEFER_WNORM: the per-group partial sums wait in `partials`
  bool wn_pending = false; const float* wn_params = nullptr; float* wn_grads = nullptr;
  // ... and, with it, the weight gradients of the plain convs around a gated stack (first conv, head): their planes stay in
  // `scratch` / the caller's `saved` until the group call
  bool pw_pending = false; int pw_B = 0, pw_T = 0; const uint16_t* pw_a = nullptr; const uint16_t* pw_b = nullptr;
  // grown on demand
  float* partials = nullptr; long long partial_cap = 0;
  float* scratch = nullptr; long long scratch_cap = 0;
  std::vector<long long> pt_per_utt;  // per entry partial floats per utterance (0 if shared)
  // weight-gradient partial sums: two regions with their own slot counts - the gated blocks'
  // convs (utterance groups, stack_wgrad_kernel) and everything else (chunk groups, table kernel)
  long long pt_floats_stack = 0, pt_floats_gen = 0;
  std::vector<ConvEntry> abs_ents;  // the uploaded table (absolute partial offsets, slot counts)
  int Gs = 0, Gg = 0, cpg_gen = 1;
  int L = 0;
  int idx_first = -1, idx_last1 = -1, idx_last2 = -1;
  std::vector<int> idx_conv, idx_aux, idx_out, idx_skip, idx_plain;
  StackLayer* d_layers = nullptr;  // fused-forward layer table (kinds 0/1)
  StackBLayer* d_blayers = nullptr;  // fused data-gradient layer table
  StackWLayer* d_wlayers = nullptr;  // fused weight-gradient layer table (partial offsets for wl_G groups)
  int wl_G = 0;
  // fused plain-conv chains (pstack_kernels.hip): device layer tables, rebuilt when the batch shape changes
  PsLayer* d_ps = nullptr;   // [4][PS_MAXL]: kind 2: forward, backward; gated: first fwd | head fwd | head bwd | first bwd
  PwLayer* d_pw = nullptr;   // [PS_MAXL] weight-gradient table
  long long ps_N = -1; int ps_Gg = -1;
  // optional side stream for the weight gradients (crk_net_set_wgrad_stream): they only read planes the
  // data-gradient chain has finished writing, so they overlap the next stack's chain on the main stream
  hipStream_t wg_stream = nullptr;
  hipEvent_t ev_chain = nullptr, ev_wg = nullptr;
  bool wg_pending = false;
  // Device tables depend on the batch shape (plane offsets are multiples of N = B*T, partial-sum offsets of the slot counts
  // Gs / Gg).  Every shape gets tables of its own that live as long as the net: d_ents / d_ps / d_pw / d_wlayers above are
  // the CURRENT shape's (switching is a host-side pointer swap), so a captured HIP graph - which holds the pointers of the
  // shape it was captured with - keeps seeing that shape's tables whatever ran in between (a short last batch of an
  // epoch, a dev batch).  Buffers that grow are retired, not freed, for the same reason.  Growth is bounded by the number of
  // DISTINCT (B, T) a run feeds a net: training has one (batch_len is fixed, dataset.py crops / pads to it), decoding one per
  // flag (batch_len = longest utterance); a table set is a few KB, so nothing is evicted - a captured graph may hold any of them.
  struct EntSet { int Gs, Gg; ConvEntry* d; std::vector<ConvEntry> abs; };
  struct PsSet { long long N; int Gs, Gg; PsLayer* d_ps; PwLayer* d_pw; };
  struct WlSet { int G, Gg; StackWLayer* d; };
  std::vector<EntSet> ent_sets; std::vector<PsSet> ps_sets; std::vector<WlSet> wl_sets;
  std::vector<void*> retired;  // outgrown partial-sum / scratch buffers (freed with the net)
  // How each recent forward laid out the planes in the caller's `saved` workspace (keyed by its address; the last 32 calls):
  // mode 0 plain bf16, 1 split operands with hi + lo planes (CRK_FLAG_PRECISE), 2 split-operand forward that saved what a
  // PLAIN backward reads (CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN); x3f: the channel-split split-operand forward wrote them.
  // crk_net_backward checks its flags against the tag instead of trusting the caller to pair the two calls.
  struct FwdTag { const float* saved; int B, T; unsigned char mode; bool x3f; };
  FwdTag fwd_tags[32]; int fwd_tag_next = 0; int fwd_tag_count = 0;
  // deferred plain-conv weight gradients: the launch parameters of the shape they were deferred for
  PwP pw_params; int pw_nw = 0, pw_max_wa = 0, pw_max_wb = 0, pw_max_tiles = 0; double pw_flops = 0.0;
  const ConvEntry* wn_ents = nullptr;  // table of the shape the pending weight-norm backward belongs to
  std::vector<WgradP> jobs;  // weight-gradient problems queued by the running backward
  WgradP* d_jobs = nullptr;
  // pinned upload ring for the job table (a slot is reused only after its copy completed)
  WgradP* h_slot[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t slot_ev[4];
  int slot_next = 0;
};

static int pad16(int c) { return round_up(c, 16); }
static int pad32(int c) { return round_up(c, 32); }

static void add_conv(Net* n, int role, int layer, int cout, int cin, int k, int dil, bool bias) {
  ConvEntry e;
  memset(&e, 0, sizeof(e));
  e.cout = cout; e.cin = cin; e.k = k;
  if (bias) { e.off_b = n->n_params; n->n_params += cout; } else e.off_b = -1;
  e.off_g = n->n_params; n->n_params += cout;
  e.off_v = n->n_params; n->n_params += (long long)cout * cin * k;
  e.norm_off = n->norm_elems; n->norm_elems += cout;
  e.bw_off = -1; e.fr_off = -1; e.fr_mode = 0; e.bfr_off = -1; e.bfr_mode = 0;
  n->ents.push_back(e);
  n->meta.push_back({role, layer, dil});
}

static long long alloc_w(Net* n, long long elems) {
  long long o = n->wprep_elems;
  n->wprep_elems += (elems + 7) & ~7ll;  // keep 16-byte alignment of every plane
  return o;
}
static long long alloc_pt(Net* n, long long floats_per_group, bool stack) {
  long long& tot = stack ? n->pt_floats_stack : n->pt_floats_gen;
  const long long o = tot;
  tot += floats_per_group;
  return o;
}

static int upload_entries(Net* n, int Gs, int Gg);
extern "C" void* crk_net_create(const crk_net_desc* desc) {
  AllocScope may_allocate;
  if (!desc) return nullptr;
  if (conv_kernels_init() != CRK_OK) return nullptr;
  Net* n = new Net();
  n->d = *desc;
  const crk_net_desc& d = n->d;
  if (d.kind == 0 || d.kind == 1) {
    if (d.res_ch != 64 || d.gate_ch != 128 || d.skip_ch != 64 || d.layers % d.stacks != 0 ||
        (d.kernel_size % 2 == 0 && !d.causal) || d.in_ch > 128 || d.out_ch > 128 || d.aux_ch > 128) {
      fprintf(stderr, "[crank_hip] net_create: unsupported gated-residual configuration\n");
      delete n;
      return nullptr;
    }
    n->L = d.layers;
    const int lps = d.layers / d.stacks;
    n->idx_first = (int)n->ents.size();
    add_conv(n, ROLE_FIRST, -1, 64, d.in_ch, 1, 1, true);
    for (int l = 0; l < d.layers; l++) {
      const int dil = 1 << (l % lps);
      n->idx_conv.push_back((int)n->ents.size());
      add_conv(n, ROLE_CONV, l, 128, 64, d.kernel_size, dil, d.use_bias);
      if (d.aux_ch > 0) {
        n->idx_aux.push_back((int)n->ents.size());
        add_conv(n, ROLE_AUX, l, 128, d.aux_ch, 1, 1, false);
      }
      n->idx_out.push_back((int)n->ents.size());
      add_conv(n, ROLE_OUT, l, 64, 64, 1, 1, d.use_bias);
      n->idx_skip.push_back((int)n->ents.size());
      add_conv(n, ROLE_SKIP, l, 64, 64, 1, 1, d.use_bias);
    }
    n->idx_last1 = (int)n->ents.size();
    add_conv(n, ROLE_LAST1, -1, 64, 64, 1, 1, true);
    n->idx_last2 = (int)n->ents.size();
    add_conv(n, ROLE_LAST2, -1, d.out_ch, 64, 1, 1, true);
  } else if (d.kind == 2) {
    if (d.kernel_size % 2 == 0 || d.conv_ch > 128 || d.in_ch > 128 || d.out_ch > 128 || d.layers < 1) {
      delete n;
      return nullptr;
    }
    n->L = d.layers;
    int cin = d.in_ch;
    for (int i = 0; i < d.layers - 1; i++) {
      const int dil = (i == 0) ? 1 : i;  // dilation_factor == 1
      n->idx_plain.push_back((int)n->ents.size());
      add_conv(n, ROLE_PLAIN, i, d.conv_ch, cin, d.kernel_size, dil, d.use_bias);
      cin = d.conv_ch;
    }
    n->idx_plain.push_back((int)n->ents.size());
    add_conv(n, ROLE_PLAIN, d.layers - 1, d.out_ch, cin, d.kernel_size, 1, d.use_bias);
  } else {
    delete n;
    return nullptr;
  }
  // ---- operand-plane and partial layouts ----
  for (size_t i = 0; i < n->ents.size(); i++) {
    ConvEntry& e = n->ents[i];
    const ConvMeta& m = n->meta[i];
    e.pt_scale = 1.f;
    switch (m.role) {
      case ROLE_OUT: {
        // combined [out|skip] planes are laid out when the OUT entry is visited
        ConvEntry& sk = n->ents[i + 1];
        e.fw_rows = sk.fw_rows = 128; e.fw_kp = sk.fw_kp = 64;
        e.fw_off = sk.fw_off = alloc_w(n, 128 * 64);
        e.fw_row0 = 0; sk.fw_row0 = 64;
        e.bw_rows = sk.bw_rows = 64; e.bw_kp = sk.bw_kp = 128;
        e.bw_off = sk.bw_off = alloc_w(n, 64 * 128);
        e.bw_col0 = 0; sk.bw_col0 = 64;
        e.pt_rows = sk.pt_rows = 128; e.pt_cx = sk.pt_cx = 64; e.pt_taps = sk.pt_taps = 1;
        e.pt_groups = sk.pt_groups = 1;
        e.pt_off = sk.pt_off = alloc_pt(n, 128 * 64, true);
        e.pb_off = sk.pb_off = alloc_pt(n, 128, true);
        e.pt_row0 = 0; sk.pt_row0 = 64;
        e.pt_scale = 0.70710678118654752440f; sk.pt_scale = 1.f;
        e.fr_off = sk.fr_off = alloc_w(n, 4 * 4 * 64 * 8); e.fr_mode = 3; sk.fr_mode = 4;
        e.bfr_off = sk.bfr_off = alloc_w(n, 64 * 128);
        break;
      }
      case ROLE_SKIP:
        break;  // filled with its OUT sibling
      default: {
        e.fw_rows = pad32(e.cout); e.fw_kp = pad16(e.cin); e.fw_row0 = 0;
        // the conditioning 1x1 of a gated block is consumed by the fused forward kernel as one more
        // 128 x 64 weight chunk (zero columns beyond aux_ch), exactly like a tap
        if (m.role == ROLE_AUX && e.cin <= 64) e.fw_kp = 64;
        e.fw_off = alloc_w(n, (long long)e.k * e.fw_rows * e.fw_kp);
        e.bw_rows = pad32(e.cin); e.bw_kp = pad16(e.cout); e.bw_col0 = 0;
        e.bw_off = alloc_w(n, (long long)e.k * e.bw_rows * e.bw_kp);
        e.pt_rows = e.cout; e.pt_row0 = 0; e.pt_cx = e.cin; e.pt_taps = e.k;
        e.pt_groups = (m.role == ROLE_CONV || m.role == ROLE_AUX) ? 1 : 0;
        e.pt_off = alloc_pt(n, (long long)e.k * e.cout * e.cin, e.pt_groups != 0);
        e.pb_off = alloc_pt(n, e.cout, e.pt_groups != 0);
        if (m.role == ROLE_CONV && e.cin == 64 && e.cout == 128) { e.fr_off = alloc_w(n, (long long)e.k * 4 * 4 * 64 * 8); e.fr_mode = 1; }
        if (m.role == ROLE_AUX && e.cin <= 64 && e.cout == 128) { e.fr_off = alloc_w(n, 4 * 4 * 64 * 8); e.fr_mode = 2; }
        if ((m.role == ROLE_FIRST || m.role == ROLE_LAST1 || m.role == ROLE_LAST2) && e.k == 1) {
          e.fr_off = alloc_w(n, (long long)(e.fw_rows >> 5) * (e.fw_kp >> 4) * 64 * 8); e.fr_mode = 5;
        }
        if (d.kind == 0 && (m.role == ROLE_CONV || m.role == ROLE_AUX || m.role == ROLE_FIRST || m.role == ROLE_LAST1 ||
                            m.role == ROLE_LAST2))
          e.bfr_off = alloc_w(n, (long long)e.k * e.bw_rows * e.bw_kp);
        // the discriminator's data-gradient chain runs channel-split too (round 4): its transposed tap weights in A-fragment order
        if (d.kind == 1 && m.role == ROLE_CONV && e.cin == 64 && e.cout == 128) e.bfr_off = alloc_w(n, (long long)e.k * e.bw_rows * e.bw_kp);
        if (m.role == ROLE_PLAIN) {  // kind-2 chains: both layouts in fragment order, a tile's fragments in one run
          e.fr_off = alloc_w(n, (long long)e.k * e.fw_rows * e.fw_kp); e.fr_mode = 6;
          e.bfr_off = alloc_w(n, (long long)e.k * e.bw_rows * e.bw_kp); e.bfr_mode = 1;
        }
        break;
      }
    }
  }
  bool ok = true;
  ok = ok && NET_MALLOC(&n->whi, sizeof(uint16_t) * n->wprep_elems) == hipSuccess;
  ok = ok && NET_MALLOC(&n->wlo, sizeof(uint16_t) * n->wprep_elems) == hipSuccess;
  ok = ok && NET_MALLOC(&n->norms, sizeof(float) * n->norm_elems) == hipSuccess;
  ok = ok && hipMemset(n->whi, 0, sizeof(uint16_t) * n->wprep_elems) == hipSuccess;
  ok = ok && hipMemset(n->wlo, 0, sizeof(uint16_t) * n->wprep_elems) == hipSuccess;
  if (ok && (d.kind == 0 || d.kind == 1)) {
    std::vector<StackLayer> lt(n->L);
    for (int l = 0; l < n->L; l++) {
      const ConvEntry& ec = n->ents[n->idx_conv[l]];
      const ConvEntry& eo = n->ents[n->idx_out[l]];
      const ConvEntry& es = n->ents[n->idx_skip[l]];
      StackLayer& y = lt[l];
      y.w_conv = ec.fw_off; y.w_os = eo.fw_off;
      y.w_aux = d.aux_ch > 0 ? n->ents[n->idx_aux[l]].fw_off : 0;
      y.f_conv = ec.fr_off; y.f_os = eo.fr_off;
      y.f_aux = d.aux_ch > 0 ? n->ents[n->idx_aux[l]].fr_off : -1;
      y.b_conv = ec.off_b; y.b_out = eo.off_b; y.b_skip = es.off_b;
      y.dil = n->meta[n->idx_conv[l]].dilation;
      y.off0 = d.causal ? -(ec.k - 1) * y.dil : -((ec.k - 1) / 2) * y.dil;
    }
    ok = NET_MALLOC(&n->d_layers, sizeof(StackLayer) * n->L) == hipSuccess &&
         hipMemcpy(n->d_layers, lt.data(), sizeof(StackLayer) * n->L, hipMemcpyHostToDevice) == hipSuccess;
    std::vector<StackBLayer> bt(n->L);
    for (int l = 0; l < n->L; l++) {
      const ConvEntry& ec = n->ents[n->idx_conv[l]];
      const ConvEntry& eo = n->ents[n->idx_out[l]];
      StackBLayer& y = bt[l];
      y.w_conv = ec.bw_off; y.w_os = eo.bw_off;
      y.w_aux = d.aux_ch > 0 ? n->ents[n->idx_aux[l]].bw_off : 0;
      y.dil = n->meta[n->idx_conv[l]].dilation;
      const int off0 = d.causal ? -(ec.k - 1) * y.dil : -((ec.k - 1) / 2) * y.dil;
      y.off0 = -off0 - (ec.k - 1) * y.dil;
      y.f_conv = ec.bfr_off; y.f_os = eo.bfr_off;
      y.f_aux = d.aux_ch > 0 ? n->ents[n->idx_aux[l]].bfr_off : -1;
    }
    ok = ok && NET_MALLOC(&n->d_blayers, sizeof(StackBLayer) * n->L) == hipSuccess &&
         hipMemcpy(n->d_blayers, bt.data(), sizeof(StackBLayer) * n->L, hipMemcpyHostToDevice) == hipSuccess;
  }
  ok = ok && upload_entries(n, 1, 1) == CRK_OK;  // (the table weight preparation reads; batch shapes get their own in crk_net_reserve)
  if (!ok) {
    fprintf(stderr, "[crank_hip] net_create: device allocation failed\n");
    delete n;
    return nullptr;
  }
  return n;
}

// partial offsets are per group; the device table needs absolute offsets for the slot counts of a
// given batch shape: [stack region: entry block x Gs slots][generic region: entry block x Gg slots]
static int upload_entries(Net* n, int Gs, int Gg) {
  for (auto& es : n->ent_sets)
    if (es.Gs == Gs && es.Gg == Gg) { n->d_ents = es.d; n->abs_ents = es.abs; n->Gs = Gs; n->Gg = Gg; return CRK_OK; }
  if (!g_may_alloc) return not_reserved("conv-entry table");
  Net::EntSet es; es.Gs = Gs; es.Gg = Gg; es.d = nullptr;
  es.abs = n->ents;
  for (auto& e : es.abs) {
    const bool stack = e.pt_groups != 0;
    const long long base = stack ? 0 : n->pt_floats_stack * Gs;
    const int G = stack ? Gs : Gg;
    e.pt_off = base + e.pt_off * G; e.pb_off = base + e.pb_off * G; e.pt_groups = G;
  }
  if (NET_MALLOC(&es.d, sizeof(ConvEntry) * es.abs.size()) != hipSuccess) return CRK_ERR_HIP;
  if (hipMemcpy(es.d, es.abs.data(), sizeof(ConvEntry) * es.abs.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(es.d);
    return CRK_ERR_HIP;  // (e.g. a new batch shape first seen inside a stream capture: run it eagerly once)
  }
  n->ent_sets.push_back(es);
  n->d_ents = es.d; n->abs_ents = es.abs; n->Gs = Gs; n->Gg = Gg;
  return CRK_OK;
}

extern "C" void crk_net_destroy(void* h) {
  Net* n = (Net*)h;
  if (!n) return;
  for (auto& es : n->ent_sets) (void)hipFree(es.d);
  for (auto& ps : n->ps_sets) { (void)hipFree(ps.d_ps); (void)hipFree(ps.d_pw); }
  for (auto& ws : n->wl_sets) (void)hipFree(ws.d);
  for (void* q : n->retired) (void)hipFree(q);
  (void)hipFree(n->whi); (void)hipFree(n->wlo); (void)hipFree(n->norms);
  if (n->ev_chain) { (void)hipEventDestroy(n->ev_chain); (void)hipEventDestroy(n->ev_wg); }
  for (int k = 0; k < 4; k++) if (n->h_slot[k]) { (void)hipHostFree(n->h_slot[k]); (void)hipEventDestroy(n->slot_ev[k]); }
  (void)hipFree(n->partials); (void)hipFree(n->scratch); (void)hipFree(n->d_jobs); (void)hipFree(n->d_layers); (void)hipFree(n->d_blayers);
  delete n;
}

extern "C" long long crk_net_param_count(void* h) { return ((Net*)h)->n_params; }
extern "C" int crk_net_conv_count(void* h) { return (int)((Net*)h)->ents.size(); }
// out[0..8] = cout, cin, k, off_bias, off_g, off_v, dilation, role, layer
extern "C" int crk_net_conv_info(void* h, int i, long long* out) {
  Net* n = (Net*)h;
  if (i < 0 || i >= (int)n->ents.size()) return CRK_ERR_ARG;
  const ConvEntry& e = n->ents[i];
  out[0] = e.cout; out[1] = e.cin; out[2] = e.k; out[3] = e.off_b; out[4] = e.off_g; out[5] = e.off_v;
  out[6] = n->meta[i].dilation; out[7] = n->meta[i].role; out[8] = n->meta[i].layer;
  return CRK_OK;
}

static int stack_aux_pad(const Net* n) { return n->d.aux_ch > 0 ? n->ents[n->idx_aux[0]].fw_kp : 16; }
// ---- workspace layouts -------------------------------------------------------------------
// `saved` (caller-owned, forward -> backward):
//   kind 2: [fp32 pre-activations H_0..H_{L-2} (per-layer fallback only)] then bf16 operand planes
//           O_i [N, kp_i] of every conv, hi block then lo block
//   gated : fp32 planes X | TA | SB | Z | SKIP | H1 (TA, SB, Z, H1: per-layer fallback only), then bf16:
//           Xb Zb Tb Sg (block input, z, tanh, sigmoid; hi[L] lo[L], [N,64] each), Cb_hi Cb_lo ([N,aux_pad]), F_hi F_lo (first-conv input [N,kpF]),
//           head_hi = S|H1 ([N,64] each), head_lo
#define PS_MAXL 16
static long long saved_f32_floats(const Net* n, long long N) {
  if (n->d.kind == 2) return (long long)(n->L - 1) * N * n->d.conv_ch;
  return (long long)(4 * n->L + 2) * N * 64;
}
static long long plain_planes_w(const Net* n) {  // sum of the operand-plane widths of a kind-2 net
  long long w = 0;
  for (int i = 0; i < n->L; i++) w += n->ents[n->idx_plain[i]].fw_kp;
  return w;
}
static long long plain_gplanes_w(const Net* n) {  // ... of its output-gradient planes
  long long w = 0;
  for (int i = 0; i < n->L; i++) w += n->ents[n->idx_plain[i]].bw_kp;
  return w;
}
struct GatedB16 {  // element offsets inside the gated forward bf16 region
  long long xb_hi, xb_lo, zb_hi, zb_lo, tb_hi, tb_lo, sg_hi, sg_lo, cb_hi, cb_lo, f_hi, f_lo, head_hi, head_lo, total;
};
static long long ts_plane_stride(long long N) { return ((N + 31) & ~31ll) * 64; }
static GatedB16 gated_b16(const Net* n, long long N) {
  GatedB16 g;
  const long long P = N * 64, LP = (long long)n->L * P, ca = N * stack_aux_pad(n), kf = N * n->ents[n->idx_first].fw_kp;
  // the tanh / sigmoid planes may be kept in blocks of 32 frames (StackP::ts_stride): room for a last partial block
  const long long LPt = (long long)n->L * ts_plane_stride(N);
  g.xb_hi = 0; g.xb_lo = LP; g.zb_hi = 2 * LP; g.zb_lo = 3 * LP;
  g.tb_hi = 4 * LP; g.tb_lo = g.tb_hi + LPt; g.sg_hi = g.tb_lo + LPt; g.sg_lo = g.sg_hi + LPt;
  g.cb_hi = g.sg_lo + LPt; g.cb_lo = g.cb_hi + ca;
  g.f_hi = g.cb_lo + ca; g.f_lo = g.f_hi + kf;
  g.head_hi = g.f_lo + kf; g.head_lo = g.head_hi + 2 * P;
  g.total = g.head_lo + 2 * P;
  return g;
}
static long long saved_floats(const Net* n, long long N) {
  if (n->d.kind == 2) return saved_f32_floats(n, N) + N * plain_planes_w(n);  // 2 planes (hi, lo) x 2 bytes
  return saved_f32_floats(n, N) + (gated_b16(n, N).total + 1) / 2;
}
extern "C" long long crk_net_saved_bytes(void* h, int B, int T) { return saved_floats((Net*)h, (long long)B * T) * 4; }

extern "C" int crk_net_set_wgrad_stream(void* h, void* stream) {
  Net* n = (Net*)h;
  if (!n) return CRK_ERR_ARG;
  if (n->wg_pending && n->ev_wg && hipEventSynchronize(n->ev_wg) != hipSuccess) return CRK_ERR_HIP;
  n->wg_pending = false;
  n->wg_stream = (hipStream_t)stream;
  if (stream && !n->ev_chain) {
    if (hipEventCreateWithFlags(&n->ev_chain, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&n->ev_wg, hipEventDisableTiming) != hipSuccess) return CRK_ERR_HIP;
  }
  return CRK_OK;
}
// the main stream must not touch what earlier side-stream weight gradients of this handle still use
// (scratch planes, partial sums, the weight-norm factors) before they are done
static int wait_side_work(Net* n, hipStream_t s) {
  if (n->wg_pending) {
    if (hipStreamWaitEvent(s, n->ev_wg, 0) != hipSuccess) return CRK_ERR_HIP;
    n->wg_pending = false;
  }
  return CRK_OK;
}
// stream for the weight gradients of the running backward: the side stream, ordered after everything
// the main stream has enqueued so far
static int fork_wgrad(Net* n, hipStream_t s, hipStream_t* ws) {
  *ws = s;
  if (!n->wg_stream || n->wg_stream == s) return CRK_OK;
  if (hipEventRecord(n->ev_chain, s) != hipSuccess || hipStreamWaitEvent(n->wg_stream, n->ev_chain, 0) != hipSuccess)
    return CRK_ERR_HIP;
  *ws = n->wg_stream;
  return CRK_OK;
}
static int join_wgrad(Net* n, hipStream_t s, hipStream_t ws) {
  if (ws == s) return CRK_OK;
  if (hipEventRecord(n->ev_wg, ws) != hipSuccess) return CRK_ERR_HIP;
  n->wg_pending = true;
  return CRK_OK;
}

static int net_nmax(const Net* n) {  // largest cin * k of the net's convs
  int m = 1;
  for (const auto& e : n->ents) if (e.cin * e.k > m) m = e.cin * e.k;
  return m;
}
static int ensure_prepared(Net* n, const float* params, unsigned long long version, hipStream_t s) {
  if (n->prepared_version == version && n->prepared_params == params) return CRK_OK;
  { int rc = wait_side_work(n, s); if (rc) return rc; }
  int rc = launch_weight_prep(n->d_ents, (int)n->ents.size(), net_nmax(n), params, n->whi, n->wlo, n->norms, s);
  if (rc) return rc;
  n->prepared_version = version;
  n->prepared_params = params;
  return CRK_OK;
}

// partial sums -> dg / dv / dbias: now, or (deferred) when the caller finishes all its nets with crk_nets_wnorm_bwd
static int finish_wnorm(Net* n, const float* params, float* grads, bool defer, hipStream_t s) {
  if (defer) { n->wn_pending = true; n->wn_params = params; n->wn_grads = grads; n->wn_ents = n->d_ents; return CRK_OK; }
  return launch_wnorm_bwd(n->d_ents, (int)n->ents.size(), params, grads, n->partials, n->norms, s);
}
static int flush_pending_plain_wgrad(Net* n, hipStream_t s);
static int flush_pending_wnorm(Net* n, hipStream_t s) {
  if (!n->wn_pending) return CRK_OK;
  { int rc = flush_pending_plain_wgrad(n, s); if (rc) return rc; }
  n->wn_pending = false;
  { int rc = wait_side_work(n, s); if (rc) return rc; }  // the partial sums may still be in flight on the side stream
  return launch_wnorm_bwd(n->wn_ents ? n->wn_ents : n->d_ents, (int)n->ents.size(), n->wn_params, n->wn_grads, n->partials, n->norms, s);
}

static ConvP base_conv(const Net* n, int B, int T) {
  ConvP p;
  memset(&p, 0, sizeof(p));
  p.scaleA = p.scaleB = 1.f; p.out_scale = 1.f; p.res_scale = 1.f;
  p.slope = n->d.slope;
  p.B = B; p.T = T; p.tiles_per_utt = ceil_div(T, CRK_TM);
  p.ktaps = 1; p.dil = 1; p.off0 = 0;
  static int dbg = -1;
  if (dbg < 0) dbg = 0;
  p.dbg = dbg;
  return p;
}
static void set_fw_weights(const Net* n, ConvP& p, const ConvEntry& e, const float* params) {
  p.w_hi = n->whi + e.fw_off; p.w_lo = n->wlo + e.fw_off;
  p.cin = e.cin; p.cin_pad = e.fw_kp; p.cout = e.cout; p.cout_pad = e.fw_rows;
  p.bias = e.off_b >= 0 ? params + e.off_b : nullptr;
}
static void set_bw_weights(const Net* n, ConvP& p, const ConvEntry& e) {
  // data gradient: "cin" = forward cout, "cout" = forward cin
  p.w_hi = n->whi + e.bw_off; p.w_lo = n->wlo + e.bw_off;
  p.cin = e.cout; p.cin_pad = e.bw_kp; p.cout = e.cin; p.cout_pad = e.bw_rows;
  p.bias = nullptr;
}
static int fwd_off0(const Net* n, int k, int dil) { return n->d.causal ? -(k - 1) * dil : -((k - 1) / 2) * dil; }

static unsigned long long layer_seed(unsigned long long seed, int l) { return seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(l + 1); }

#define RUN(x) do { int rc_ = (x); if (rc_ != CRK_OK) return rc_; } while (0)

static int conv_go(ConvP& p, int mode, bool precise, hipStream_t s) {
  conv_fill_lds(p, mode, precise);
  return launch_conv(p, mode, precise, s);
}

// flags bit0: precise (bf16x3 split) arithmetic
// ---- fused plain-conv chains (pstack_kernels.hip): layer tables --------------------------------
static PsLayer ps_layer_fwd(const Net* n, int ei, int epi) {
  const ConvEntry& e = n->ents[ei];
  PsLayer y; memset(&y, 0, sizeof(y));
  y.w_off = e.fw_off; y.b_off = e.off_b; y.rows = e.cout; y.rows_pad = e.fw_rows; y.kp = e.fw_kp;
  y.k = e.k; y.dil = n->meta[ei].dilation; y.off0 = -((e.k - 1) / 2) * y.dil; y.epi = epi;
  y.f_off = e.fr_mode == 6 ? e.fr_off : -1;
  return y;
}
static PsLayer ps_layer_bwd(const Net* n, int ei, int epi) {  // the conv transposed: data gradient
  const ConvEntry& e = n->ents[ei];
  PsLayer y; memset(&y, 0, sizeof(y));
  y.w_off = e.bw_off; y.b_off = -1; y.rows = e.cin; y.rows_pad = e.bw_rows; y.kp = e.bw_kp;
  y.k = e.k; y.dil = n->meta[ei].dilation; y.off0 = ((e.k - 1) / 2) * y.dil - (e.k - 1) * y.dil; y.epi = epi;
  y.f_off = e.bfr_mode == 1 ? e.bfr_off : -1;
  return y;
}
static PwLayer pw_layer(const Net* n, int ei, long long a_hi, long long a_lo, long long b_hi, long long b_lo) {
  const ConvEntry& e = n->ents[ei];
  const ConvEntry& a = n->abs_ents[ei];
  PwLayer y; memset(&y, 0, sizeof(y));
  y.a_hi = a_hi; y.a_lo = a_lo; y.b_hi = b_hi; y.b_lo = b_lo;
  y.wa = e.bw_kp; y.wb = e.fw_kp; y.ca = e.cout; y.cb = e.cin;
  y.k = e.k; y.dil = n->meta[ei].dilation; y.off0 = -((e.k - 1) / 2) * y.dil;
  y.pt = a.pt_off; y.pb = e.off_b >= 0 ? a.pb_off : -1;
  return y;
}
static bool pw_ok(const Net* n, int ei) {
  const ConvEntry& e = n->ents[ei];
  return pstack_wgrad_supported(e.cout, e.cin, e.bw_kp, e.fw_kp, e.k, n->meta[ei].dilation) != 0;
}
struct GatedS16 {  // element offsets inside the gated backward bf16 region (n->scratch)
  long long gb_hi, gb_lo, dxb_hi, dxb_lo, dsb_hi, dsb_lo, hb_hi, hb_lo, total;
};
static GatedS16 gated_s16(const Net* n, long long N) {
  GatedS16 g;
  const long long P = N * 64, L = n->L, hb = N * n->ents[n->idx_last2].bw_kp + P;
  g.gb_hi = 0; g.gb_lo = 2 * L * P; g.dxb_hi = 4 * L * P; g.dxb_lo = g.dxb_hi + (L + 1) * P;
  g.dsb_hi = g.dxb_lo + (L + 1) * P; g.dsb_lo = g.dsb_hi + P;
  g.hb_hi = g.dsb_lo + P; g.hb_lo = g.hb_hi + hb; g.total = g.hb_lo + hb;
  return g;
}
// host copies of the chain tables: [0] first/forward, [1] head forward / backward, [2] head backward, [3] first backward
struct PsTables { PsLayer t[4][PS_MAXL]; int L[4]; PwLayer w[PS_MAXL]; int nw; int max_wa, max_wb; double wflops_per_frame; };
static void ps_build(const Net* n, long long N, PsTables& T) {
  memset(&T, 0, sizeof(T));
  const crk_net_desc& d = n->d;
  if (d.kind == 2) {
    const int L = n->L;
    long long ooff[PS_MAXL], goff[PS_MAXL], ow = 0, gw = 0;
    for (int i = 0; i < L; i++) {
      ooff[i] = N * ow; goff[i] = N * gw;
      ow += n->ents[n->idx_plain[i]].fw_kp; gw += n->ents[n->idx_plain[i]].bw_kp;
    }
    for (int i = 0; i < L; i++) {
      T.t[0][i] = ps_layer_fwd(n, n->idx_plain[i], i < L - 1 ? ACT_LRELU : 0);
      T.t[0][i].save_plane = ooff[i];
      const int j = L - 1 - i;  // position in the backward chain
      T.t[1][j] = ps_layer_bwd(n, n->idx_plain[i], i > 0 ? 2 + ACT_LRELU : 0);
      T.t[1][j].save_plane = goff[i];
      if (i > 0) { T.t[1][j].mask_plane = ooff[i]; T.t[1][j].mask_w = n->ents[n->idx_plain[i]].fw_kp; }
      T.w[i] = pw_layer(n, n->idx_plain[i], goff[i], N * gw + goff[i], ooff[i], N * ow + ooff[i]);
    }
    T.L[0] = T.L[1] = L; T.nw = L;
  } else {
    const int hact = d.kind == 1 ? ACT_LRELU : ACT_RELU;
    const long long P = N * 64, kpY = n->ents[n->idx_last2].bw_kp;
    const GatedB16 gf = gated_b16(n, N);
    const GatedS16 gs = gated_s16(n, N);
    T.t[0][0] = ps_layer_fwd(n, n->idx_first, d.kind == 1 ? ACT_LRELU : 0); T.L[0] = 1;
    T.t[1][0] = ps_layer_fwd(n, n->idx_last1, hact); T.t[1][0].save_plane = 0;
    T.t[1][1] = ps_layer_fwd(n, n->idx_last2, 0); T.t[1][1].save_plane = P; T.L[1] = 2;
    T.t[2][0] = ps_layer_bwd(n, n->idx_last2, 2 + hact); T.t[2][0].mask_plane = P; T.t[2][0].mask_w = 64; T.t[2][0].save_plane = 0;
    T.t[2][1] = ps_layer_bwd(n, n->idx_last1, 2 + hact); T.t[2][1].mask_plane = 0; T.t[2][1].mask_w = 64; T.t[2][1].save_plane = N * kpY;
    T.L[2] = 2;
    T.t[3][0] = ps_layer_bwd(n, n->idx_first, 0); T.L[3] = 1;
    T.w[0] = pw_layer(n, n->idx_first, gs.dxb_hi, gs.dxb_lo, gf.f_hi, gf.f_lo);
    T.w[1] = pw_layer(n, n->idx_last1, gs.hb_hi + N * kpY, gs.hb_lo + N * kpY, gf.head_hi, gf.head_lo);
    T.w[2] = pw_layer(n, n->idx_last2, gs.hb_hi, gs.hb_lo, gf.head_hi + P, gf.head_lo + P);
    T.nw = 3;
  }
  for (int i = 0; i < T.nw; i++) {
    if (T.w[i].wa > T.max_wa) T.max_wa = T.w[i].wa;
    if (T.w[i].wb > T.max_wb) T.max_wb = T.w[i].wb;
    T.wflops_per_frame += 2.0 * T.w[i].ca * T.w[i].cb * T.w[i].k;
  }
}
static double ps_flops(const PsLayer* t, int L, long long N) {
  double f = 0.0;
  for (int i = 0; i < L; i++) f += 2.0 * (double)N * t[i].rows * t[i].kp * t[i].k;
  return f;
}
// upload the tables for this batch shape / slot counts (cached)
static int ps_upload(Net* n, long long N) {
  if (n->ps_N == N && n->ps_Gg == n->Gg * 1000 + n->Gs) return CRK_OK;
  for (auto& ps : n->ps_sets)
    if (ps.N == N && ps.Gs == n->Gs && ps.Gg == n->Gg) {
      n->d_ps = ps.d_ps; n->d_pw = ps.d_pw; n->ps_N = N; n->ps_Gg = n->Gg * 1000 + n->Gs;
      return CRK_OK;
    }
  if (!g_may_alloc) return not_reserved("plain-chain tables");
  PsTables T;
  ps_build(n, N, T);
  Net::PsSet ps; ps.N = N; ps.Gs = n->Gs; ps.Gg = n->Gg; ps.d_ps = nullptr; ps.d_pw = nullptr;
  if (NET_MALLOC(&ps.d_ps, sizeof(PsLayer) * 4 * PS_MAXL) != hipSuccess) return CRK_ERR_HIP;
  if (NET_MALLOC(&ps.d_pw, sizeof(PwLayer) * PS_MAXL) != hipSuccess) { (void)hipFree(ps.d_ps); return CRK_ERR_HIP; }
  if (hipMemcpy(ps.d_ps, T.t, sizeof(PsLayer) * 4 * PS_MAXL, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ps.d_pw, T.w, sizeof(PwLayer) * PS_MAXL, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(ps.d_ps); (void)hipFree(ps.d_pw);
    return CRK_ERR_HIP;
  }
  n->ps_sets.push_back(ps);
  n->d_ps = ps.d_ps; n->d_pw = ps.d_pw;
  n->ps_N = N; n->ps_Gg = n->Gg * 1000 + n->Gs;
  return CRK_OK;
}
static int ps_chain_version() {
  static int v = -1;
  if (v < 0) v = crk_sw().ps_v;
  return v;
}
static PsP ps_base(const Net* n, int B, int T, const float* params) {
  PsP p; memset(&p, 0, sizeof(p));
  p.in_scale = 1.f; p.out_scale = 1.f; p.params = params; p.whi = n->whi; p.wlo = n->wlo;
  p.B = B; p.T = T; p.slope = n->d.slope;
  return p;
}
// can this kind-2 net / the first conv and head of this gated net run through the fused chains?
static bool plain_chains_ok(const Net* n, int B, int T, bool precise) {
  PsTables Tb;
  ps_build(n, (long long)B * T, Tb);
  const int nchains = n->d.kind == 2 ? 2 : 4;
  for (int c = 0; c < nchains; c++) {
    if (Tb.L[c] > PS_MAXL) return false;
    PsP p = ps_base(n, B, T, nullptr);
    p.L = Tb.L[c];
    if (pstack_plan(p, Tb.t[c], precise) != CRK_OK) return false;
  }
  if (n->d.kind == 2) { for (int i = 0; i < n->L; i++) if (!pw_ok(n, n->idx_plain[i])) return false; }
  else if (!pw_ok(n, n->idx_first) || !pw_ok(n, n->idx_last1) || !pw_ok(n, n->idx_last2)) return false;
  return true;
}

// One decision for the whole stack and shape: forward, data-gradient chain and weight gradient
// run fused together or not at all (the fused kernels exchange bf16 planes the generic kernels
// do not produce).  CRK_NO_FUSE=1 selects the per-layer kernels (debugging / A-B timing).
static void stack_halo(const Net* n, int* hl, int* hr, int* max_off, int* max_dil) {
  *hl = *hr = *max_off = 0; *max_dil = 1;
  for (int l = 0; l < n->L; l++) {
    const int dil = n->meta[n->idx_conv[l]].dilation;
    const int o0 = n->d.causal ? -(n->d.kernel_size - 1) * dil : -((n->d.kernel_size - 1) / 2) * dil;
    const int o1 = o0 + (n->d.kernel_size - 1) * dil;
    *hl += -o0; *hr += o1;
    if (-o0 > *max_off) *max_off = -o0;
    if (o1 > *max_off) *max_off = o1;
    if (dil > *max_dil) *max_dil = dil;
  }
}
static bool stack_fused(const Net* n, int B, int T, bool precise) {
  static int no_fuse = -1;
  if (no_fuse < 0) no_fuse = crk_sw().no_fuse;
  if (no_fuse || n->L > PS_MAXL) return false;
  if (n->d.kind == 2) return plain_chains_ok(n, B, T, precise);
  int hl, hr, mo, md;
  stack_halo(n, &hl, &hr, &mo, &md);
  StackP sp; memset(&sp, 0, sizeof(sp));
  sp.B = B; sp.T = T; sp.L = n->L; sp.ktaps = n->d.kernel_size; sp.hl = hl; sp.hr = hr; sp.max_off = mo;
  sp.aux_ch = n->d.aux_ch > 0 ? n->d.aux_ch : 0; sp.aux_pad = stack_aux_pad(n);
  StackBP bp; memset(&bp, 0, sizeof(bp));
  bp.B = B; bp.T = T; bp.L = n->L; bp.ktaps = n->d.kernel_size; bp.hl = hr; bp.hr = hl; bp.max_off = mo;
  bp.aux_ch = sp.aux_ch;
  return stack_fwd_plan(sp, precise) == CRK_OK && stack_bwd_plan(bp, precise) == CRK_OK &&
         stack_wgrad_supported(n->d.kernel_size, md, sp.aux_ch) && plain_chains_ok(n, B, T, precise);
}

// Generator stacks in plain bf16: forward and data-gradient chain both run channel-split (stack2_kernels.hip,
// stack2b_kernels.hip) and exchange the tanh / sigmoid planes in the lane-record layout.  ONE predicate for both calls - it
// depends on the net, the batch shape and process-wide switches only, never on a call's pointers - so a backward always reads
// the layout its forward wrote.  (Misaligned tensors then fail loudly in the call instead of taking another path.)
static bool gen_split_path(const Net* n, int B, int T, bool precise) {
  const crk_net_desc& d = n->d;
  static int sk_v = -1, skb_v = -1;
  if (sk_v < 0) sk_v = crk_sw().sk_v;
  if (skb_v < 0) skb_v = crk_sw().skb_v;
  if (precise || d.kind != 0 || d.dropout != 0.f || sk_v != 2 || skb_v != 2) return false;
  if (d.in_ch % 8 || d.out_ch % 8 || d.out_ch > 128) return false;
  if (!stack_fused(n, B, T, precise)) return false;
  const ConvEntry& ef = n->ents[n->idx_first];
  const ConvEntry& e1 = n->ents[n->idx_last1];
  const ConvEntry& e2 = n->ents[n->idx_last2];
  if (ef.fr_off < 0 || e1.fr_off < 0 || e2.fr_off < 0 || ef.bfr_off < 0 || e1.bfr_off < 0 || e2.bfr_off < 0 ||
      n->ents[n->idx_conv[0]].bfr_off < 0) return false;
  int hl, hr, mo, md;
  stack_halo(n, &hl, &hr, &mo, &md);
  StackP sp; memset(&sp, 0, sizeof(sp));
  sp.B = B; sp.T = T; sp.L = n->L; sp.ktaps = d.kernel_size; sp.hl = hl; sp.hr = hr; sp.max_off = mo;
  sp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0; sp.aux_pad = stack_aux_pad(n);
  sp.x_in = reinterpret_cast<const float*>(n);  // (any non-null value: folded, the plan sizes the first conv's input tile)
  sp.in_ch = d.in_ch; sp.kp_first = ef.fw_kp;
  StackBP bp; memset(&bp, 0, sizeof(bp));
  bp.B = B; bp.T = T; bp.L = n->L; bp.ktaps = d.kernel_size; bp.hl = hr; bp.hr = hl; bp.max_off = mo; bp.aux_ch = sp.aux_ch;
  return stack2_fwd_plan(sp) == CRK_OK && stack2_bwd_plan(bp) == CRK_OK;
}

// bf16x3f (forward in split-operand arithmetic, CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN; backward in plain bf16,
// CRK_FLAG_FWD_PRECISE): generator stacks run the channel-split split-operand forward (stack2x_kernels.hip), which leaves the
// hi planes in the plain path's layout, and the plain path's backward kernels behind it.  ONE predicate for both calls, like
// gen_split_path.  CRK_S2X=0: the round-4 pairing (frame-split stack_fwd_kernel<PRECISE>, frame-split chain on row planes).
static bool gen_x3f_path(const Net* n, int B, int T) {
  static int s2x = -1;
  if (s2x < 0) s2x = crk_sw().s2x;
  if (!s2x || !gen_split_path(n, B, T, false)) return false;
  const crk_net_desc& d = n->d;
  int hl, hr, mo, md;
  stack_halo(n, &hl, &hr, &mo, &md);
  StackP sp; memset(&sp, 0, sizeof(sp));
  sp.B = B; sp.T = T; sp.L = n->L; sp.ktaps = d.kernel_size; sp.hl = hl; sp.hr = hr; sp.max_off = mo;
  sp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0; sp.aux_pad = stack_aux_pad(n);
  sp.x_in = reinterpret_cast<const float*>(n);  // (any non-null value: folded; stack2_fwd_plan, called first, sizes the input tile)
  sp.in_ch = d.in_ch; sp.kp_first = n->ents[n->idx_first].fw_kp;
  return stack2x_fwd_plan(sp) == CRK_OK;
}

// The discriminator (kind 1, no conditioning) in plain bf16, dropout or not: the forward's gated blocks (stack2_fwd_kernel, not
// folded: first conv and head keep their own launches) and the data-gradient chain (stack2_bwd_kernel<.., FOLD = false>) run
// channel-split and exchange the gate planes in the lane-record layout.  One predicate for both calls, like gen_split_path.
// CRK_DISC_SPLIT=0: the round-3 pairing (channel-split forward, frame-split chain, row-layout planes).
static bool disc_split_path(const Net* n, int B, int T, bool precise) {
  const crk_net_desc& d = n->d;
  static int sk_v = -1, skb_v = -1, dsp = -1;
  if (sk_v < 0) sk_v = crk_sw().sk_v;
  if (skb_v < 0) skb_v = crk_sw().skb_v;
  if (dsp < 0) dsp = crk_sw().disc_split;
  if (precise || d.kind != 1 || d.aux_ch > 0 || sk_v != 2 || skb_v != 2 || !dsp) return false;
  if (!stack_fused(n, B, T, precise)) return false;
  if (n->ents[n->idx_conv[0]].bfr_off < 0 || n->ents[n->idx_out[0]].bfr_off < 0 || n->ents[n->idx_conv[0]].fr_off < 0 ||
      n->ents[n->idx_out[0]].fr_off < 0) return false;
  int hl, hr, mo, md;
  stack_halo(n, &hl, &hr, &mo, &md);
  StackP sp; memset(&sp, 0, sizeof(sp));
  sp.B = B; sp.T = T; sp.L = n->L; sp.ktaps = d.kernel_size; sp.hl = hl; sp.hr = hr; sp.max_off = mo; sp.drop_p = d.dropout;
  StackBP bp; memset(&bp, 0, sizeof(bp));
  bp.B = B; bp.T = T; bp.L = n->L; bp.ktaps = d.kernel_size; bp.hl = hr; bp.hr = hl; bp.max_off = mo;
  return stack2_fwd_plan(sp) == CRK_OK && stack2_bwd_plan(bp) == CRK_OK;
}

static void tag_forward(Net* n, const float* saved, int B, int T, int flags, bool x3f) {
  if (!saved || (flags & CRK_FLAG_NO_SAVE)) return;
  for (int i = 0; i < n->fwd_tag_count; i++)
    if (n->fwd_tags[i].saved == saved) { n->fwd_tags[i] = {saved, B, T, (unsigned char)((flags & CRK_FLAG_PRECISE) ? ((flags & CRK_FLAG_BWD_PLAIN) ? 2 : 1) : 0), x3f}; return; }
  n->fwd_tags[n->fwd_tag_next] = {saved, B, T, (unsigned char)((flags & CRK_FLAG_PRECISE) ? ((flags & CRK_FLAG_BWD_PLAIN) ? 2 : 1) : 0), x3f};
  n->fwd_tag_next = (n->fwd_tag_next + 1) % 32;
  if (n->fwd_tag_count < 32) n->fwd_tag_count++;
}
// CRK_ERR_ARG when the backward's flags do not describe the forward that filled `saved` (an unknown workspace - evicted from
// the ring, or written through another handle - passes: the caller's pairing is all there is then)
static int check_forward_tag(const Net* n, const float* saved, int B, int T, int flags, bool expects_x3f) {
  for (int i = 0; i < n->fwd_tag_count; i++) {
    const Net::FwdTag& t = n->fwd_tags[i];
    if (t.saved != saved) continue;
    const int want = (flags & CRK_FLAG_PRECISE) ? 1 : ((flags & CRK_FLAG_FWD_PRECISE) ? 2 : 0);
    if (t.B != B || t.T != T || t.mode != want || (want == 2 && t.x3f != expects_x3f)) {
      fprintf(stderr, "[crank_hip] crk_net_backward: flags 0x%x (plane layout %d%s, B %d, T %d) do not match the forward that wrote this "
                      "workspace (layout %d%s, B %d, T %d): CRK_FLAG_PRECISE pairs with CRK_FLAG_PRECISE, CRK_FLAG_PRECISE | CRK_FLAG_BWD_PLAIN "
                      "with CRK_FLAG_FWD_PRECISE, plain with plain\n", flags, want, expects_x3f ? " x3f" : "", B, T, t.mode, t.x3f ? " x3f" : "", t.B, t.T);
      return CRK_ERR_ARG;
    }
    return CRK_OK;
  }
  return CRK_OK;
}
// what a batch shape needs: scratch / partial-sum floats and the slot counts of the two weight-gradient regions
struct ShapeNeed { long long need_s, need_p; int Gs, Gg, cpg; };
static ShapeNeed shape_need(const Net* n, int B, int T);
static int select_shape(Net* n, const ShapeNeed& q);
extern "C" int crk_net_forward(void* h, const float* params, unsigned long long version, const float* x, int ldx,
                               const float* c, int ldc, float* y, int ldy, float* saved, int B, int T, int flags,
                               unsigned long long seed, void* stream) {
  Net* n = (Net*)h;
  if (!n || !params || !x || !y || B <= 0 || T <= 0) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool precise = flags & CRK_FLAG_PRECISE;
  // the call's dropout seed: a value, or (CRK_FLAG_SEED_ON_DEVICE) the address of one in device memory
  const unsigned long long* seed_ptr = (flags & CRK_FLAG_SEED_ON_DEVICE) ? reinterpret_cast<const unsigned long long*>((uintptr_t)seed) : nullptr;
  const unsigned long long seed_val = (flags & CRK_FLAG_SEED_ON_DEVICE) ? 0ull : seed;
  const crk_net_desc& d = n->d;
  RUN(ensure_prepared(n, params, version, s));
  RUN(select_shape(n, shape_need(n, B, T)));  // this shape's tables (a pointer swap; CRK_ERR_ARG: the shape was not reserved)
  tag_forward(n, saved, B, T, flags, precise && (flags & CRK_FLAG_BWD_PLAIN) && n->d.kind == 0 && gen_x3f_path(n, B, T));
  const long long N = (long long)B * T;
  if (d.kind == 2 && stack_fused(n, B, T, precise)) {
    // the whole stack in one launch; every conv's input operand is kept as a bf16 plane
    if (!saved && !(flags & CRK_FLAG_NO_SAVE)) return CRK_ERR_ARG;
    RUN(ps_upload(n, N));
    PsTables Tb;
    ps_build(n, N, Tb);
    PsP p = ps_base(n, B, T, params);
    p.x = x; p.ldx = ldx; p.cin = d.in_ch; p.y = y; p.ldy = ldy;
    if (!(flags & CRK_FLAG_NO_SAVE)) {
      p.save_hi = reinterpret_cast<uint16_t*>(saved + saved_f32_floats(n, N));
      p.save_lo = p.save_hi + N * plain_planes_w(n);
    }
    p.layers = n->d_ps; p.L = Tb.L[0];
    if (!precise && ps_chain_version() == 2) {  // channel-split chain (pstack2_kernels.hip); CRK_PS_V=1: the frame-split one
      PsP q = p;
      if (pstack2_plan(q, Tb.t[0]) == CRK_OK) return launch_pstack2(q, ps_flops(Tb.t[0], Tb.L[0], N), s);
    }
    if (precise && (flags & CRK_FLAG_BWD_PLAIN) && ps_chain_version() == 2) {  // bf16x3f: split-operand forward, hi planes only
      static int s2x = -1;
      if (s2x < 0) s2x = crk_sw().s2x;
      PsP q = p;
      q.save_lo = nullptr;
      if (s2x && pstack2x_plan(q, Tb.t[0]) == CRK_OK) return launch_pstack2x(q, ps_flops(Tb.t[0], Tb.L[0], N), s);
    }
    RUN(pstack_plan(p, Tb.t[0], precise));
    return launch_pstack(p, precise, ps_flops(Tb.t[0], Tb.L[0], N), s);
  }
  if (d.kind == 2) {
    const int L = n->L;
    if (L > 1 && !saved) return CRK_ERR_ARG;
    const float* in = x; int ldin = ldx;
    for (int i = 0; i < L; i++) {
      const ConvEntry& e = n->ents[n->idx_plain[i]];
      const int dil = n->meta[n->idx_plain[i]].dilation;
      ConvP p = base_conv(n, B, T);
      set_fw_weights(n, p, e, params);
      p.xa = in; p.lda = ldin; p.cinA = e.cin;
      p.act_in = (i == 0) ? ACT_NONE : ACT_LRELU;
      p.ktaps = e.k; p.dil = dil; p.off0 = -((e.k - 1) / 2) * dil;
      if (i == L - 1) { p.y = y; p.ldy = ldy; }
      else { p.y = saved + (long long)i * N * d.conv_ch; p.ldy = d.conv_ch; }
      RUN(conv_go(p, MODE_PLAIN, precise, s));
      in = p.y; ldin = p.ldy;
    }
    return CRK_OK;
  }
  if (!saved || (d.aux_ch > 0 && !c)) return CRK_ERR_ARG;
  const int L = n->L;
  const long long P = N * 64;
  float* X = saved;                 // X[l], l < L
  float* TA = saved + (long long)L * P;
  float* SB = TA + (long long)L * P;
  float* Z = SB + (long long)L * P;
  float* SKIP = Z + (long long)L * P;
  float* H1 = SKIP + P;
  const int head_act = d.kind == 1 ? ACT_LRELU : ACT_RELU;
  const bool fused = stack_fused(n, B, T, precise);
  PsTables Tb;
  uint16_t* b16 = reinterpret_cast<uint16_t*>(saved + saved_f32_floats(n, N));
  const GatedB16 gf = gated_b16(n, N);
  const bool keep = !(flags & CRK_FLAG_NO_SAVE);
  // plain bf16, generator stacks: first conv, gated blocks and head in ONE launch (stack2_kernels.hip)
  bool folded = false;
  const bool x3f = precise && (flags & CRK_FLAG_BWD_PLAIN) && d.kind == 0 && gen_x3f_path(n, B, T);
  if ((fused && !precise && d.kind == 0) || x3f) {
    static int sk_v = -1;
    if (sk_v < 0) sk_v = crk_sw().sk_v;
    const ConvEntry& ef = n->ents[n->idx_first];
    const ConvEntry& e1 = n->ents[n->idx_last1];
    const ConvEntry& e2 = n->ents[n->idx_last2];
    StackP sp;
    memset(&sp, 0, sizeof(sp));
    sp.c = c; sp.ldc = ldc; sp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0; sp.aux_pad = stack_aux_pad(n);
    sp.params = params;
    if (keep) {
      sp.saved = saved;
      sp.xb_hi = b16 + gf.xb_hi; sp.zb_hi = b16 + gf.zb_hi; sp.tb_hi = b16 + gf.tb_hi; sp.sg_hi = b16 + gf.sg_hi;
      if (d.aux_ch > 0) sp.cb_hi = b16 + gf.cb_hi;
      sp.fin_hi = b16 + gf.f_hi; sp.head_hi = b16 + gf.head_hi;
    }
    sp.skip = SKIP;  // (unused by the folded kernel; a valid base for its dummy descriptors)
    sp.whi = n->whi; sp.wlo = n->wlo; sp.layers = n->d_layers;
    sp.B = B; sp.T = T; sp.L = L; sp.ktaps = d.kernel_size;
    int md;
    stack_halo(n, &sp.hl, &sp.hr, &sp.max_off, &md);
    sp.x_in = x; sp.ldx_in = ldx; sp.in_ch = d.in_ch; sp.kp_first = ef.fw_kp;
    sp.f_first = ef.fr_off; sp.b_first = ef.off_b;
    sp.f_h1 = e1.fr_off; sp.b_h1 = e1.off_b; sp.f_h2 = e2.fr_off; sp.b_h2 = e2.off_b;
    sp.y = y; sp.ldy = ldy; sp.out_ch = d.out_ch; sp.head_scale = (float)sqrt(1.0 / L);
    const bool shape_ok = (d.in_ch % 8 == 0) && (ldx % 4 == 0) && (d.out_ch % 4 == 0) && (ldy % 4 == 0) && d.out_ch <= 128 &&
                          ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)y) & 15) == 0) && ef.fr_off >= 0 && e1.fr_off >= 0 && e2.fr_off >= 0;
    const bool split = x3f || gen_split_path(n, B, T, precise);
    if (split && !shape_ok) {
      fprintf(stderr, "[crank_hip] crk_net_forward: x / y must be 16-byte aligned with row strides that are multiples of 4 floats\n");
      return CRK_ERR_ARG;
    }
    if (split) sp.ts_stride = ts_plane_stride(N);
    if (x3f) {  // split-operand arithmetic, the plain path's planes
      RUN(stack2x_fwd_plan(sp));
      RUN(launch_stack2x_fwd(sp, s));
      folded = true;
    } else if (sk_v == 2 && shape_ok && d.dropout == 0.f && stack2_fwd_plan(sp) == CRK_OK) {
      RUN(launch_stack2_fwd(sp, s));
      folded = true;
    } else if (split) return CRK_ERR_UNSUPPORTED;  // (cannot happen: the predicate implies the plan)
  }
  if (folded) return CRK_OK;
  if (fused) {  // first conv (1x1; kind 1: + LeakyReLU) -> X_0, its input kept as a bf16 plane
    RUN(ps_upload(n, N));
    ps_build(n, N, Tb);
    PsP p = ps_base(n, B, T, params);
    p.x = x; p.ldx = ldx; p.cin = d.in_ch; p.y = X; p.ldy = 64;
    if (keep) { p.save_hi = b16 + gf.f_hi; p.save_lo = b16 + gf.f_lo; }
    p.layers = n->d_ps; p.L = 1;
    RUN(pstack_plan(p, Tb.t[0], precise));
    RUN(launch_pstack(p, precise, ps_flops(Tb.t[0], 1, N), s));
  } else
  {  // first conv (kind 1: followed by LeakyReLU)
    const ConvEntry& e = n->ents[n->idx_first];
    ConvP p = base_conv(n, B, T);
    set_fw_weights(n, p, e, params);
    p.xa = x; p.lda = ldx; p.cinA = e.cin;
    p.y = X; p.ldy = 64; p.act_out = d.kind == 1 ? ACT_LRELU : ACT_NONE;
    RUN(conv_go(p, MODE_PLAIN, precise, s));
  }
  if (fused) {
    StackP sp;
    memset(&sp, 0, sizeof(sp));
    sp.x0 = X; sp.c = c; sp.ldc = ldc; sp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0;
    sp.aux_pad = stack_aux_pad(n);
    sp.skip = SKIP; sp.params = params;
    if (keep) {
      sp.saved = saved;
      sp.xb_hi = b16 + gf.xb_hi; sp.xb_lo = b16 + gf.xb_lo; sp.zb_hi = b16 + gf.zb_hi; sp.zb_lo = b16 + gf.zb_lo;
      sp.tb_hi = b16 + gf.tb_hi; sp.tb_lo = b16 + gf.tb_lo; sp.sg_hi = b16 + gf.sg_hi; sp.sg_lo = b16 + gf.sg_lo;
      if (d.aux_ch > 0) { sp.cb_hi = b16 + gf.cb_hi; sp.cb_lo = b16 + gf.cb_lo; }
    }
    sp.whi = n->whi; sp.wlo = n->wlo; sp.layers = n->d_layers;
    sp.B = B; sp.T = T; sp.L = L; sp.ktaps = d.kernel_size;
    int md;
    stack_halo(n, &sp.hl, &sp.hr, &sp.max_off, &md);
    if (d.dropout > 0.f) { sp.drop_p = d.dropout; sp.drop_seed = seed_val; sp.drop_seed_ptr = seed_ptr; }
    // plain bf16: the channel-split kernel (stack2_kernels.hip); bf16x3 and CRK_SK_V=1: the frame-split one
    static int sk_v = -1;
    if (sk_v < 0) sk_v = crk_sw().sk_v;
    if (disc_split_path(n, B, T, precise)) sp.ts_stride = ts_plane_stride(N);  // (its data-gradient chain reads lane records)
    if (!precise && sk_v != 1 && stack2_fwd_plan(sp) == CRK_OK) {
      RUN(launch_stack2_fwd(sp, s));
    } else {
      if (sp.ts_stride) return CRK_ERR_UNSUPPORTED;  // (cannot happen: the predicate implies the plan)
      RUN(stack_fwd_plan(sp, precise));
      RUN(launch_stack_fwd(sp, precise, s));
    }
  }
  for (int l = 0; l < L && !fused; l++) {
    const ConvEntry& ec = n->ents[n->idx_conv[l]];
    const ConvEntry& eo = n->ents[n->idx_out[l]];
    const ConvEntry& es = n->ents[n->idx_skip[l]];
    const int dil = n->meta[n->idx_conv[l]].dilation;
    ConvP p = base_conv(n, B, T);
    set_fw_weights(n, p, ec, params);
    p.xa = X + l * P; p.lda = 64; p.cinA = 64;
    p.ktaps = ec.k; p.dil = dil; p.off0 = fwd_off0(n, ec.k, dil);
    if (d.dropout > 0.f) { p.drop_p = d.dropout; p.drop_seed = layer_seed(seed_val, l); p.drop_seed_ptr = seed_ptr; }
    if (d.aux_ch > 0) {
      const ConvEntry& ea = n->ents[n->idx_aux[l]];
      p.xc = c; p.ldc = ldc; p.cinC = ea.cin; p.cinC_pad = ea.fw_kp;
      p.wc_hi = n->whi + ea.fw_off; p.wc_lo = n->wlo + ea.fw_off;
    }
    p.w2_hi = n->whi + eo.fw_off; p.w2_lo = n->wlo + eo.fw_off;
    p.bias2a = eo.off_b >= 0 ? params + eo.off_b : nullptr;
    p.bias2b = es.off_b >= 0 ? params + es.off_b : nullptr;
    p.y = (l < L - 1) ? X + (l + 1) * P : nullptr; p.ldy = 64;
    p.skip = SKIP; p.skip_init = (l == 0);
    p.sv_ta = TA + l * P; p.sv_sb = SB + l * P; p.sv_z = Z + l * P;
    RUN(conv_go(p, MODE_RESFWD, precise, s));
  }
  if (fused) {  // head: act(skips * sqrt(1/L)) -> 1x1 -> act -> 1x1, one launch; both operands kept as bf16 planes
    PsP p = ps_base(n, B, T, params);
    p.x = SKIP; p.ldx = 64; p.cin = 64; p.in_scale = (float)sqrt(1.0 / L); p.in_act = head_act;
    p.y = y; p.ldy = ldy;
    if (keep) { p.save_hi = b16 + gf.head_hi; p.save_lo = b16 + gf.head_lo; }
    p.layers = n->d_ps + PS_MAXL; p.L = 2;
    RUN(pstack_plan(p, Tb.t[1], precise));
    RUN(launch_pstack(p, precise, ps_flops(Tb.t[1], 2, N), s));
  } else
  {  // head: act(skips * sqrt(1/L)) -> 1x1 -> act -> 1x1
    const ConvEntry& e1 = n->ents[n->idx_last1];
    ConvP p = base_conv(n, B, T);
    set_fw_weights(n, p, e1, params);
    p.xa = SKIP; p.lda = 64; p.cinA = 64; p.scaleA = (float)sqrt(1.0 / L); p.act_in = head_act;
    p.y = H1; p.ldy = 64;
    RUN(conv_go(p, MODE_PLAIN, precise, s));
    const ConvEntry& e2 = n->ents[n->idx_last2];
    ConvP q = base_conv(n, B, T);
    set_fw_weights(n, q, e2, params);
    q.xa = H1; q.lda = 64; q.cinA = 64; q.act_in = head_act;
    q.y = y; q.ldy = ldy;
    RUN(conv_go(q, MODE_PLAIN, precise, s));
  }
  return CRK_OK;
}

// Weight-gradient groups of a net's "stack region" (the gated blocks' launch is one workgroup per (group, block); the
// partial sums are read back `groups` times by the weight-norm backward): runs of 64-frame chunks, utterance after
// utterance.  32 groups of whole utterances by default; a gated stack of few blocks gets as many groups as fill the 256
// compute units with its (group, block) workgroups - 6 blocks x 32 groups are 192 workgroups of 16 chunks each (B = 64,
// T = 500), 6 x 40 are 240 of 13.  A stack of 8 blocks keeps its 32 groups of two utterances: same sums, bit for bit.
static int stack_cpg(const Net* n, int B, int T) {
  const int ncpu = (T + 63) / 64;
  const int groups = crk_sw().wg_groups;
  const int gsz = (B + groups - 1) / groups;  // utterances per group
  int cpg = gsz * ncpu;
  if (crk_sw().wg_fill && n->d.kind != 2 && n->L > 0 && 256 / n->L > groups) {
    const int fill = 256 / n->L;
    const int c = (B * ncpu + fill - 1) / fill;
    if (c >= 1 && c < cpg) cpg = c;
  }
  return cpg < 1 ? 1 : cpg;
}
static int stack_groups(const Net* n, int B, int T) {
  const int total = B * ((T + 63) / 64), cpg = stack_cpg(n, B, T);
  return (total + cpg - 1) / cpg;
}

static ShapeNeed shape_need(const Net* n, int B, int T) {
  ShapeNeed q;
  const long long N = (long long)B * T;
  const long long cw = n->d.conv_ch > n->d.out_ch ? n->d.conv_ch : n->d.out_ch;
  // every layer keeps its own gradient buffers: the weight gradients of the whole stack
  // run as ONE launch after the data-gradient chain
  // gated stacks: dS | dH1 | dX_l (L+1) | dG_l (2L) fp32 planes, then the bf16 planes of the fused chain:
  // dGb_hi[L] dGb_lo[L] ([N,128]), dXb_hi[L+1] dXb_lo[L+1], dSb_hi dSb_lo ([N,64])
  // (+ head: dy and dH1 bf16 planes);  kind 2: per-layer fp32 gradients (fallback) + bf16 output-gradient planes
  q.need_s = n->d.kind == 2 ? (long long)n->L * N * cw + N * plain_gplanes_w(n)
                            : N * 64 * (3LL * n->L + 3) + (gated_s16(n, N).total + 1) / 2;
  q.Gs = stack_groups(n, B, T);
  // generic convs: runs of 64-frame chunks, at most 32 groups: short runs = many workgroups hide the latency of the
  // table kernel's load -> MFMA chain, but every group is one more pass of the weight-norm backward over the
  // partial sums and one more set-up / partial-sum write-out (12 % + 21 % of a workgroup's life at 8 chunks per group).
  // Round 2, one-deep prefetch: 128 groups 2.14 ms/step, 64 groups 2.10, 51 groups 2.12.  Round 6, chunks requested two
  // ahead: 64 groups (8 chunks each) 1.490 ms/step, 43 groups 1.496, 32 groups (16 each) 1.481, 16 groups 1.512
  // (profiles/round6_c_envs.txt).
  // ... and a chain of few convs (the speaker-adversarial net: 3) keeps 64 groups: its launch is (groups x convs) workgroups,
  // 96 of them for 256 CUs at 32 groups (22.0 -> 30.5 us, profiles/round6_c_kernel_stats.csv)
  const int total_chunks = B * ((T + 63) / 64);
  const int groups = (n->d.kind == 2 && n->ents.size() <= 4) ? 64 : 32;
  q.cpg = (total_chunks + groups - 1) / groups;
  if (crk_sw().wg_cpg > 0) q.cpg = crk_sw().wg_cpg;
  q.Gg = (total_chunks + q.cpg - 1) / q.cpg;
  q.need_p = n->pt_floats_stack * q.Gs + n->pt_floats_gen * q.Gg;
  return q;
}
// the current tables become those of batch shape (B, T); nothing is allocated unless inside crk_net_reserve
static int select_shape(Net* n, const ShapeNeed& q) {
  n->cpg_gen = q.cpg;
  if (n->Gs != q.Gs || n->Gg != q.Gg) { RUN(upload_entries(n, q.Gs, q.Gg)); n->wl_G = 0; }
  return CRK_OK;
}
static int ensure_bwd_buffers(Net* n, int B, int T) {
  const ShapeNeed q = shape_need(n, B, T);
  if ((q.need_s > n->scratch_cap || q.need_p > n->partial_cap || !n->d_jobs) && !g_may_alloc) return not_reserved("crk_net_backward");
  if (q.need_s > n->scratch_cap) {
    if (n->scratch) n->retired.push_back(n->scratch);  // (a captured graph may still hold the pointer)
    n->scratch = nullptr; n->scratch_cap = 0;
    if (NET_MALLOC(&n->scratch, q.need_s * 4) != hipSuccess) return CRK_ERR_HIP;
    n->scratch_cap = q.need_s;
  }
  if (q.need_p > n->partial_cap) {
    if (n->partials) n->retired.push_back(n->partials);
    n->partials = nullptr; n->partial_cap = 0;
    if (NET_MALLOC(&n->partials, q.need_p * 4) != hipSuccess) return CRK_ERR_HIP;
    n->partial_cap = q.need_p;
  }
  RUN(select_shape(n, q));
  if (!n->d_jobs) {
    if (NET_MALLOC(&n->d_jobs, sizeof(WgradP) * 256) != hipSuccess) return CRK_ERR_HIP;
  }
  return CRK_OK;
}

// fused weight-gradient layer table of a gated stack for G utterance groups (built once per count, kept: see Net::ent_sets)
static int ensure_wl_table(Net* n, int G) {
  if (n->wl_G == G) return CRK_OK;
  const crk_net_desc& d = n->d;
  const int L = n->L;
  StackWLayer* found = nullptr;
  for (auto& ws : n->wl_sets) if (ws.G == G && ws.Gg == n->Gg) found = ws.d;
  if (!found) {
    if (!g_may_alloc) return not_reserved("weight-gradient layer table");
    std::vector<StackWLayer> wt(L);
    for (int l = 0; l < L; l++) {
      const ConvEntry& ec = n->ents[n->idx_conv[l]];
      const ConvEntry& eo = n->ents[n->idx_out[l]];
      StackWLayer& y = wt[l];
      const ConvEntry& ac = n->abs_ents[n->idx_conv[l]];
      const ConvEntry& ao = n->abs_ents[n->idx_out[l]];
      y.pt_conv = ac.pt_off; y.pb_conv = ec.off_b >= 0 ? ac.pb_off : -1;
      y.pt_os = ao.pt_off; y.pb_os = eo.off_b >= 0 ? ao.pb_off : -1;
      y.pt_aux = d.aux_ch > 0 ? n->abs_ents[n->idx_aux[l]].pt_off : 0;
      y.dil = n->meta[n->idx_conv[l]].dilation;
      y.off0 = fwd_off0(n, ec.k, y.dil);
    }
    Net::WlSet ws; ws.G = G; ws.Gg = n->Gg; ws.d = nullptr;
    if (NET_MALLOC(&ws.d, sizeof(StackWLayer) * L) != hipSuccess) return CRK_ERR_HIP;
    if (hipMemcpy(ws.d, wt.data(), sizeof(StackWLayer) * L, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(ws.d); return CRK_ERR_HIP; }
    n->wl_sets.push_back(ws);
    found = ws.d;
  }
  n->d_wlayers = found;
  n->wl_G = G;
  return CRK_OK;
}

extern "C" int crk_net_reserve(void* h, int B, int T) {
  Net* n = (Net*)h;
  if (!n || B <= 0 || T <= 0) return CRK_ERR_ARG;
  AllocScope may_allocate;
  RUN(ensure_bwd_buffers(n, B, T));
  if (n->d.kind != 2) RUN(ensure_wl_table(n, stack_groups(n, B, T)));
  if (n->L <= PS_MAXL) RUN(ps_upload(n, (long long)B * T));
  return CRK_OK;
}
extern "C" long long crk_net_scratch_bytes(void* h, int B, int T) {
  Net* n = (Net*)h;
  if (!n || B <= 0 || T <= 0) return -1;
  const ShapeNeed q = shape_need(n, B, T);
  return (q.need_s + q.need_p) * 4;
}
extern "C" long long crk_debug_alloc_count(void) { return g_net_allocs; }
// which kernel generation the compute entry points pick for a batch shape (the predicates they share): bit 0 the generator
// stack runs channel-split in plain bf16 (stack2_fwd_kernel / stack2_bwd_kernel), bit 1 its bf16x3f forward runs on the
// channel-split split-operand kernel (stack2x_fwd_kernel), bit 2 the discriminator's blocks and chain run channel-split,
// bit 3 the net is a chain of plain convs that runs fused (pstack kernels).  Every fallback computes the same values, only
// slower: a test pins the bits at the benchmark shape so that a plan that starts failing does not pass as a timing.
extern "C" int crk_debug_net_paths(void* h, int B, int T) {
  Net* n = (Net*)h;
  if (!n || B <= 0 || T <= 0) return -1;
  int r = 0;
  if (n->d.kind == 0 && gen_split_path(n, B, T, false)) r |= 1;
  if (n->d.kind == 0 && gen_x3f_path(n, B, T)) r |= 2;
  if (n->d.kind == 1 && disc_split_path(n, B, T, false)) r |= 4;
  if (n->d.kind == 2 && stack_fused(n, B, T, false)) r |= 8;
  return r;
}

static WgradP base_wgrad(const Net* n, int B, int T) {
  WgradP w;
  memset(&w, 0, sizeof(w));
  w.sa1 = w.sa2 = w.sx = 1.f; w.slope = n->d.slope;
  w.B = B; w.T = T; w.ktaps = 1; w.dil = 1; w.off0 = 0;
  w.dbg = 0;
  return w;
}
// partial-sum slots of conv entry ei: pointers into the partial block, chunks per group, group count
static void wgrad_slots(const Net* n, int ei, int B, int T, WgradP& w) {
  const ConvEntry& a = n->abs_ents[ei];
  const bool stack = n->ents[ei].pt_groups != 0;
  w.partial = n->partials + a.pt_off;
  w.bias_partial = a.off_b >= 0 ? n->partials + a.pb_off : nullptr;
  w.ngroups = a.pt_groups;
  w.cpg = stack ? stack_cpg(n, B, T) : n->cpg_gen;
}
// queue one weight-gradient problem; launched with the rest of the stack's by wgrad_flush
static int wgrad_go(Net* n, WgradP& w, bool precise) {
  w.ca_pad = pad32(w.ca); w.cx_pad = pad32(w.cx); w.cc_pad = w.has_aux ? pad32(w.cc) : 0;
  return wgrad_expand(w, precise, n->jobs);
}
static int wgrad_flush(Net* n, int B, int T, bool precise, hipStream_t s) {
  if (n->jobs.empty()) return CRK_OK;
  if (n->jobs.size() > 256) return CRK_ERR_UNSUPPORTED;
  const int k = n->slot_next;
  n->slot_next = (k + 1) & 3;
  if (!n->h_slot[k]) {
    if (hipHostMalloc((void**)&n->h_slot[k], sizeof(WgradP) * 256, hipHostMallocDefault) != hipSuccess) return CRK_ERR_HIP;
    if (hipEventCreateWithFlags(&n->slot_ev[k], hipEventDisableTiming) != hipSuccess) return CRK_ERR_HIP;
  } else if (hipEventSynchronize(n->slot_ev[k]) != hipSuccess) {
    return CRK_ERR_HIP;
  }
  memcpy(n->h_slot[k], n->jobs.data(), sizeof(WgradP) * n->jobs.size());
  if (hipMemcpyAsync(n->d_jobs, n->h_slot[k], sizeof(WgradP) * n->jobs.size(), hipMemcpyHostToDevice, s) != hipSuccess)
    return CRK_ERR_HIP;
  if (hipEventRecord(n->slot_ev[k], s) != hipSuccess) return CRK_ERR_HIP;
  int mg = 0;
  for (const auto& j : n->jobs) if (j.ngroups > mg) mg = j.ngroups;
  int rc = launch_wgrad_table(n->d_jobs, n->jobs, B, T, mg, precise, s);
  n->jobs.clear();
  return rc;
}

// weight gradients of the plain convs of a net from the bf16 planes of its fused chains
static PwP plain_wgrad_params(Net* n, int B, int T, const uint16_t* abase, const uint16_t* bbase) {
  PwP wp; memset(&wp, 0, sizeof(wp));
  wp.layers = n->d_pw; wp.abase = abase; wp.bbase = bbase; wp.partials = n->partials;
  wp.B = B; wp.T = T; wp.cpg = n->cpg_gen; wp.G = n->Gg;
  return wp;
}
static int ps_max_tiles(const PsTables& Tb) {  // largest (tap, cin band, cout band) tile count of a conv of the table
  int mt = 0;
  for (int i = 0; i < Tb.nw; i++) {
    const int t = ((Tb.w[i].ca + 31) / 32) * ((Tb.w[i].cb + 31) / 32) * Tb.w[i].k;
    if (t > mt) mt = t;
  }
  return mt;
}
static int plain_wgrad(Net* n, int B, int T, const uint16_t* abase, const uint16_t* bbase, bool precise, hipStream_t s) {
  PsTables Tb;
  ps_build(n, (long long)B * T, Tb);
  const PwP wp = plain_wgrad_params(n, B, T, abase, bbase);
  return launch_pstack_wgrad(wp, Tb.nw, Tb.max_wa, Tb.max_wb, precise, Tb.wflops_per_frame * B * T, s, ps_max_tiles(Tb));
}
static int flush_pending_plain_wgrad(Net* n, hipStream_t s) {
  if (!n->pw_pending) return CRK_OK;
  n->pw_pending = false;
  PsTables Tb;
  ps_build(n, (long long)n->pw_B * n->pw_T, Tb);
  return launch_pstack_wgrad(n->pw_params, Tb.nw, Tb.max_wa, Tb.max_wb, false, Tb.wflops_per_frame * n->pw_B * n->pw_T, s, ps_max_tiles(Tb));
}

// flags bit0: precise; bit1: skip parameter gradients (they would be discarded);
// dx / dc may be null when the corresponding input needs no gradient.
// dx_scale multiplies the returned input gradient (gradient reversal: -lambda).
static int net_backward_impl(void* h, const float* params, unsigned long long version, float* grads, const float* x,
                             int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx,
                             float dx_scale, float* dc, int lddc, const float* saved, int B, int T, int flags,
                             unsigned long long seed, const float* dy_num, const float* dy_den, void* stream);
extern "C" int crk_net_backward(void* h, const float* params, unsigned long long version, float* grads, const float* x,
                                int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx,
                                float dx_scale, float* dc, int lddc, const float* saved, int B, int T, int flags,
                                unsigned long long seed, void* stream) {
  return net_backward_impl(h, params, version, grads, x, ldx, c, ldc, dy, lddy, dx, lddx, dx_scale, dc, lddc, saved, B, T, flags,
                           seed, nullptr, nullptr, stream);
}
// crk_net_backward of dy * (dy_num[0] / dy_den[1]), the factor read on the device by the chain's first kernel: the backward
// of a mean cross entropy on the net's output (dy = softmax - onehot, dy_num = upstream gradient, dy_den = {loss, count} as
// crk_ce_fwd leaves them) without a scaling launch in between.  Chains of plain convs only (kind 2, fused path):
// CRK_ERR_UNSUPPORTED otherwise - scale with crk_ce_bwd and call crk_net_backward.
extern "C" int crk_net_backward_scaled(void* h, const float* params, unsigned long long version, float* grads, const float* x,
                                       int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx,
                                       float dx_scale, float* dc, int lddc, const float* saved, int B, int T, int flags,
                                       unsigned long long seed, const float* dy_num, const float* dy_den, void* stream) {
  Net* n = (Net*)h;
  if (!n || !dy_num || !dy_den) return CRK_ERR_ARG;
  if (n->d.kind != 2 || !stack_fused(n, B, T, flags & 1)) return CRK_ERR_UNSUPPORTED;
  return net_backward_impl(h, params, version, grads, x, ldx, c, ldc, dy, lddy, dx, lddx, dx_scale, dc, lddc, saved, B, T, flags,
                           seed, dy_num, dy_den, stream);
}
static int net_backward_impl(void* h, const float* params, unsigned long long version, float* grads, const float* x,
                             int ldx, const float* c, int ldc, const float* dy, int lddy, float* dx, int lddx,
                             float dx_scale, float* dc, int lddc, const float* saved, int B, int T, int flags,
                             unsigned long long seed, const float* dy_num, const float* dy_den, void* stream) {
  Net* n = (Net*)h;
  if (!n || !params || !x || !dy || B <= 0 || T <= 0) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool precise = flags & CRK_FLAG_PRECISE;
  // CRK_FLAG_FWD_PRECISE: how the forward laid its planes out - unless that forward was the channel-split split-operand one
  // (generator stacks of the bf16x3f mode), which writes the plain path's planes
  const bool expects_x3f = !precise && (flags & CRK_FLAG_FWD_PRECISE) && n->d.kind == 0 && gen_x3f_path(n, B, T);
  const bool planes_precise = precise || ((flags & CRK_FLAG_FWD_PRECISE) && !expects_x3f);
  RUN(check_forward_tag(n, saved, B, T, flags, expects_x3f));
  const bool want_w = !(flags & CRK_FLAG_NO_PARAM_GRAD) && grads;
  const bool defer_wn = flags & CRK_FLAG_DEFER_WNORM;
  const unsigned long long* seed_ptr = (flags & CRK_FLAG_SEED_ON_DEVICE) ? reinterpret_cast<const unsigned long long*>((uintptr_t)seed) : nullptr;
  const unsigned long long seed_val = (flags & CRK_FLAG_SEED_ON_DEVICE) ? 0ull : seed;
  const crk_net_desc& d = n->d;
  RUN(flush_pending_wnorm(n, s));  // a second backward of this net reuses the partial-sum buffer and the gradient planes
  RUN(ensure_prepared(n, params, version, s));
  RUN(wait_side_work(n, s));
  RUN(ensure_bwd_buffers(n, B, T));
  const long long N = (long long)B * T;
  float* PT = n->partials;
  const int G = stack_groups(n, B, T);
  n->jobs.clear();

  if (d.kind == 2 && stack_fused(n, B, T, precise)) {
    // data-gradient chain in one launch (output-gradient planes kept), then all weight gradients in one
    if (!saved) return CRK_ERR_ARG;
    const long long cw = d.conv_ch > d.out_ch ? d.conv_ch : d.out_ch;
    RUN(ps_upload(n, N));
    PsTables Tb;
    ps_build(n, N, Tb);
    const uint16_t* f16 = reinterpret_cast<const uint16_t*>(saved + saved_f32_floats(n, N));
    uint16_t* g16 = reinterpret_cast<uint16_t*>(n->scratch + (long long)n->L * N * cw);
    PsP p = ps_base(n, B, T, params);
    p.x = dy; p.ldx = lddy; p.cin = d.out_ch; p.y = dx; p.ldy = lddx; p.out_scale = dx_scale;
    p.in_num = dy_num; p.in_den = dy_den;
    p.save_hi = g16; p.save_lo = g16 + N * plain_gplanes_w(n);
    p.mask_hi = f16;
    p.layers = n->d_ps + PS_MAXL; p.L = Tb.L[1];
    // nobody wants the input gradient (the classifier's input is data, the adversarial net's is detached in its own
    // update): the chain stops at the output-gradient plane of the first conv - its weight gradient needs that - and the
    // transposed first conv, the widest layer of the chain, is not computed
    if (!dx && p.L >= 2) { p.L -= 1; p.tail = 1; }
    bool done = false;
    if (!precise && ps_chain_version() == 2) {
      PsP q = p;
      if (pstack2_plan(q, Tb.t[1]) == CRK_OK) { RUN(launch_pstack2(q, ps_flops(Tb.t[1], p.L, N), s)); done = true; }
    }
    if (!done) {
      RUN(pstack_plan(p, Tb.t[1], precise));
      RUN(launch_pstack(p, precise, ps_flops(Tb.t[1], p.L, N), s));
    }
    if (want_w) {
      hipStream_t ws;
      RUN(fork_wgrad(n, s, &ws));
      RUN(plain_wgrad(n, B, T, g16, f16, precise, ws));
      RUN(finish_wnorm(n, params, grads, defer_wn, ws));
      RUN(join_wgrad(n, s, ws));
    }
    return CRK_OK;
  }
  if (d.kind == 2) {
    const int L = n->L;
    const long long cw = d.conv_ch > d.out_ch ? d.conv_ch : d.out_ch;
    const float* dcur = dy; int ldcur = lddy;
    for (int i = L - 1; i >= 0; i--) {
      const int ei = n->idx_plain[i];
      const ConvEntry& e = n->ents[ei];
      const int dil = n->meta[ei].dilation;
      const float* in = (i == 0) ? x : saved + (long long)(i - 1) * N * d.conv_ch;
      const int ldin = (i == 0) ? ldx : d.conv_ch;
      if (want_w) {
        WgradP w = base_wgrad(n, B, T);
        w.a1 = dcur; w.lda1 = ldcur; w.ca1 = e.cout; w.ca = e.cout;
        w.x = in; w.ldx = ldin; w.cx = e.cin; w.act_in = (i == 0) ? ACT_NONE : ACT_LRELU;
        w.ktaps = e.k; w.dil = dil; w.off0 = -((e.k - 1) / 2) * dil;
        wgrad_slots(n, ei, B, T, w);
        RUN(wgrad_go(n, w, precise));
      }
      if (i > 0 || dx) {
        ConvP p = base_conv(n, B, T);
        set_bw_weights(n, p, e);
        p.xa = dcur; p.lda = ldcur; p.cinA = e.cout;
        p.ktaps = e.k; p.dil = dil; p.off0 = -(-((e.k - 1) / 2) * dil) - (e.k - 1) * dil;
        if (i > 0) {
          float* out = n->scratch + (long long)i * N * cw;  // dH_{i-1}, kept for its weight gradient
          p.y = out; p.ldy = d.conv_ch;
          p.dmask = in; p.ldm = ldin; p.dmask_act = ACT_LRELU;
          RUN(conv_go(p, MODE_PLAIN, precise, s));
          dcur = out; ldcur = d.conv_ch;
        } else {
          p.y = dx; p.ldy = lddx; p.out_scale = dx_scale;
          RUN(conv_go(p, MODE_PLAIN, precise, s));
        }
      }
    }
    if (want_w) {
      RUN(wgrad_flush(n, B, T, precise, s));
      RUN(finish_wnorm(n, params, grads, defer_wn, s));
    }
    return CRK_OK;
  }

  if (!saved) return CRK_ERR_ARG;
  const int L = n->L;
  const long long P = N * 64;
  const float* X = saved;
  const float* TA = saved + (long long)L * P;
  const float* SB = TA + (long long)L * P;
  const float* Z = SB + (long long)L * P;
  const float* SKIP = Z + (long long)L * P;
  const float* H1 = SKIP + P;
  float* dS = n->scratch;
  float* dH1 = dS + P;
  float* dXall = dH1 + P;                    // dX_l at dXall + l*P, l = 0..L (dX_L is never written: it is zero)
  float* dGall = dXall + (long long)(L + 1) * P;  // dG_l [N,128] at dGall + l*2P
  const int head_act = d.kind == 1 ? ACT_LRELU : ACT_RELU;
  const float sL = (float)sqrt(1.0 / L);
  const float rs = 0.70710678118654752440f;

  const bool fused = stack_fused(n, B, T, precise);
  PsTables Tb;
  const uint16_t* f16 = reinterpret_cast<const uint16_t*>(saved + saved_f32_floats(n, N));
  uint16_t* s16 = reinterpret_cast<uint16_t*>(n->scratch + N * 64 * (3LL * L + 3));
  const GatedB16 gf = gated_b16(n, N);
  const GatedS16 gs = gated_s16(n, N);
  // plain bf16, generator stacks: the head's and the first conv's data gradients run inside the chain's launch
  bool bfold = false;
  if (fused && !precise && d.kind == 0 && d.dropout == 0.f) {
    static int sk_v = -1;
    if (sk_v < 0) sk_v = crk_sw().sk_v;
    const bool splitp = gen_split_path(n, B, T, planes_precise);
    // (the channel-split chain also takes a dy whose rows are only 4-byte aligned: a column slice of a wider gradient)
    const bool ok_y = (d.out_ch % 8 == 0) && (splitp || ((lddy % 4 == 0) && ((((uintptr_t)dy) & 15) == 0))) && ((((uintptr_t)dy) & 3) == 0);
    const bool ok_x = !dx || ((d.in_ch % 4 == 0) && (lddx % 4 == 0) && ((((uintptr_t)dx) & 15) == 0));
    bfold = sk_v == 2 && ok_y && ok_x && (stack_bwd_waves(precise) == 8 || splitp);
  }
  if (fused) { RUN(ps_upload(n, N)); ps_build(n, N, Tb); }
  if (fused && !bfold) {  // head backward: dy -> dH1 -> dS in one launch; dy and dH1 kept as bf16 planes
    PsP p = ps_base(n, B, T, params);
    p.x = dy; p.ldx = lddy; p.cin = d.out_ch; p.y = dS; p.ldy = 64; p.out_scale = sL;
    p.save_hi = s16 + gs.hb_hi; p.save_lo = s16 + gs.hb_lo;
    p.mask_hi = f16 + gf.head_hi;
    p.layers = n->d_ps + 2 * PS_MAXL; p.L = 2;
    RUN(pstack_plan(p, Tb.t[2], precise));
    RUN(launch_pstack(p, precise, ps_flops(Tb.t[2], 2, N), s));
  } else if (!fused)
  {  // head
    const ConvEntry& e2 = n->ents[n->idx_last2];
    if (want_w) {
      WgradP w = base_wgrad(n, B, T);
      w.a1 = dy; w.lda1 = lddy; w.ca1 = e2.cout; w.ca = e2.cout;
      w.x = H1; w.ldx = 64; w.cx = 64; w.act_in = head_act;
      wgrad_slots(n, n->idx_last2, B, T, w);
      RUN(wgrad_go(n, w, precise));
    }
    ConvP p = base_conv(n, B, T);
    set_bw_weights(n, p, e2);
    p.xa = dy; p.lda = lddy; p.cinA = e2.cout;
    p.y = dH1; p.ldy = 64; p.dmask = H1; p.ldm = 64; p.dmask_act = head_act;
    RUN(conv_go(p, MODE_PLAIN, precise, s));
    const ConvEntry& e1 = n->ents[n->idx_last1];
    if (want_w) {
      WgradP w = base_wgrad(n, B, T);
      w.a1 = dH1; w.lda1 = 64; w.ca1 = 64; w.ca = 64;
      w.x = SKIP; w.ldx = 64; w.cx = 64; w.sx = sL; w.act_in = head_act;
      wgrad_slots(n, n->idx_last1, B, T, w);
      RUN(wgrad_go(n, w, precise));
    }
    ConvP q = base_conv(n, B, T);
    set_bw_weights(n, q, e1);
    q.xa = dH1; q.lda = 64; q.cinA = 64;
    q.y = dS; q.ldy = 64; q.dmask = SKIP; q.ldm = 64; q.dmask_act = head_act; q.out_scale = sL;
    RUN(conv_go(q, MODE_PLAIN, precise, s));
  }
  hipStream_t ws = s;  // stream of the weight-gradient launches (the side stream once the fused chain has forked)
  const float* dxo = nullptr;  // gradient wrt the block output; the last block's x output is unused
  if (fused) {
    StackBP bp;
    memset(&bp, 0, sizeof(bp));
    bp.dS = dS; bp.saved = saved; bp.dX0 = dXall;
    bp.tb_hi = f16 + gf.tb_hi; bp.tb_lo = f16 + gf.tb_lo; bp.sg_hi = f16 + gf.sg_hi; bp.sg_lo = f16 + gf.sg_lo;
    bp.gb_hi = s16 + gs.gb_hi; bp.gb_lo = s16 + gs.gb_lo;
    bp.dxb_hi = s16 + gs.dxb_hi; bp.dxb_lo = s16 + gs.dxb_lo;
    bp.dsb_hi = s16 + gs.dsb_hi; bp.dsb_lo = s16 + gs.dsb_lo;
    bp.dc = (dc && d.aux_ch > 0) ? dc : nullptr; bp.lddc = lddc; bp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0;
    bp.whi = n->whi; bp.wlo = n->wlo; bp.layers = n->d_blayers;
    bp.B = B; bp.T = T; bp.L = L; bp.ktaps = d.kernel_size;
    int fhl, fhr, md;
    stack_halo(n, &fhl, &fhr, &bp.max_off, &md);
    bp.hl = fhr; bp.hr = fhl;  // the data gradient looks the other way
    if (d.dropout > 0.f) { bp.drop_p = d.dropout; bp.drop_seed = seed_val; bp.drop_seed_ptr = seed_ptr; }
    bp.mask_l0 = d.kind == 1; bp.slope = d.slope;
    RUN(stack_bwd_plan(bp, precise));
    bool split = false;  // the channel-split chain (stack2b_kernels.hip): folded generator stacks, plain bf16
    if (bfold && (bp.nw == 8 || gen_split_path(n, B, T, planes_precise))) {
      const ConvEntry& ef = n->ents[n->idx_first];
      const ConvEntry& e1 = n->ents[n->idx_last1];
      const ConvEntry& e2 = n->ents[n->idx_last2];
      bp.dy = dy; bp.lddy = lddy; bp.out_ch = d.out_ch; bp.kp_y = e2.bw_kp;
      bp.w_h2 = e2.bw_off; bp.w_h1 = e1.bw_off; bp.w_first = ef.bw_off;
      bp.hmask_hi = f16 + gf.head_hi; bp.hb_hi = s16 + gs.hb_hi; bp.head_scale = sL;
      bp.dx = dx; bp.lddx = lddx; bp.in_ch = d.in_ch; bp.in_rows = ef.bw_rows; bp.dx_scale = dx_scale;
      // CRK_SKB_V=1: the frame-split chain (A/B timing, the bitwise test); see gen_split_path
      bp.f_h2 = e2.bfr_off; bp.f_h1 = e1.bfr_off; bp.f_first = ef.bfr_off;
      if (gen_split_path(n, B, T, planes_precise)) {
        StackBP q = bp;
        q.ts_stride = ts_plane_stride(N);
        if (stack2_bwd_plan(q) == CRK_OK) { bp = q; split = true; }
      }
    } else
      bfold = false;
    if (!split && disc_split_path(n, B, T, planes_precise)) {  // the discriminator: the same chain without the folds
      StackBP q = bp;
      q.ts_stride = ts_plane_stride(N);
      q.dy = nullptr;
      if (stack2_bwd_plan(q) != CRK_OK) return CRK_ERR_UNSUPPORTED;  // (cannot happen: the predicate implies the plan)
      bp = q; split = true;
    }
    if (!split && gen_split_path(n, B, T, planes_precise)) {
      // the forward wrote the gate planes for the channel-split chain: nothing else can read them
      fprintf(stderr, "[crank_hip] crk_net_backward: dy / dx must be 16-byte aligned with row strides that are multiples of 4 floats\n");
      return CRK_ERR_ARG;
    }
    if (!bfold && !bp.dS) return CRK_ERR_ARG;
    // the planes the weight gradient reads after this chain as 4-frame records (StackBP::rec): the channel-split chain only
    bp.rec = (split && N % 4 == 0) ? 1 : 0;
    if (split) RUN(launch_stack2_bwd(bp, s));
    else RUN(launch_stack_bwd(bp, precise, s));
    if (want_w) {
      RUN(fork_wgrad(n, s, &ws));  // everything the weight gradients read is written by now
      // weight gradients of every block: one launch over (utterance group, block)
      RUN(ensure_wl_table(n, G));
      StackWP wp;
      memset(&wp, 0, sizeof(wp));
      wp.xb_hi = f16 + gf.xb_hi; wp.xb_lo = f16 + gf.xb_lo; wp.zb_hi = f16 + gf.zb_hi; wp.zb_lo = f16 + gf.zb_lo;
      wp.aux_pad = stack_aux_pad(n);
      if (d.aux_ch > 0) { wp.cb_hi = f16 + gf.cb_hi; wp.cb_lo = f16 + gf.cb_lo; }
      wp.gb_hi = bp.gb_hi; wp.gb_lo = bp.gb_lo; wp.dxb_hi = bp.dxb_hi; wp.dxb_lo = bp.dxb_lo;
      wp.dsb_hi = bp.dsb_hi; wp.dsb_lo = bp.dsb_lo;
      wp.layers = n->d_wlayers; wp.partials = PT;
      wp.B = B; wp.T = T; wp.L = L; wp.ktaps = d.kernel_size; wp.aux_ch = d.aux_ch > 0 ? d.aux_ch : 0;
      wp.cpg = stack_cpg(n, B, T); wp.G = G;
      wp.rec_g = bp.rec;
      RUN(launch_stack_wgrad(wp, precise, ws));
    }
    dxo = dXall;
  }
  for (int l = L - 1; l >= 0 && !fused; l--) {
    const ConvEntry& ec = n->ents[n->idx_conv[l]];
    const ConvEntry& eo = n->ents[n->idx_out[l]];
    const int dil = n->meta[n->idx_conv[l]].dilation;
    const int off0 = fwd_off0(n, ec.k, dil);
    float* dG = dGall + (long long)l * 2 * P;
    {  // gate backward: dz = [dxo*sqrt(.5) | dS] . [Wo;Ws]^T ; dG = gate'(dz)
      ConvP p = base_conv(n, B, T);
      p.w_hi = n->whi + eo.bw_off; p.w_lo = n->wlo + eo.bw_off;
      p.cin = 128; p.cin_pad = 128; p.cout = 64; p.cout_pad = 64;
      p.xa = dxo; p.lda = 64; p.cinA = 64; p.scaleA = rs;
      p.xb = dS; p.ldb = 64; p.cinB = 64;
      p.ta = TA + l * P; p.sb = SB + l * P;
      p.y = dG; p.ldy = 128;
      RUN(conv_go(p, MODE_BWDA, precise, s));
    }
    if (want_w) {
      WgradP w = base_wgrad(n, B, T);  // dilated conv (+ aux as an extra table entry)
      w.a1 = dG; w.lda1 = 128; w.ca1 = 128; w.ca = 128;
      w.x = X + l * P; w.ldx = 64; w.cx = 64;
      if (d.dropout > 0.f) { w.drop_p = d.dropout; w.drop_seed = layer_seed(seed_val, l); w.drop_seed_ptr = seed_ptr; }
      w.ktaps = ec.k; w.dil = dil; w.off0 = off0;
      wgrad_slots(n, n->idx_conv[l], B, T, w);
      if (d.aux_ch > 0) {
        const ConvEntry& ea = n->ents[n->idx_aux[l]];
        w.has_aux = 1; w.xc = c; w.ldc = ldc; w.cc = ea.cin; w.partial_aux = PT + n->abs_ents[n->idx_aux[l]].pt_off;
      }
      RUN(wgrad_go(n, w, precise));
      WgradP v = base_wgrad(n, B, T);  // 1x1 out | skip on z
      v.a1 = dxo; v.lda1 = 64; v.ca1 = 64; v.a2 = dS; v.lda2 = 64; v.ca2 = 64; v.ca = 128;
      v.x = Z + l * P; v.ldx = 64; v.cx = 64;
      wgrad_slots(n, n->idx_out[l], B, T, v);
      RUN(wgrad_go(n, v, precise));
    }
    if (dc && d.aux_ch > 0) {  // conditioning gradient, accumulated over layers
      const ConvEntry& ea = n->ents[n->idx_aux[l]];
      ConvP p = base_conv(n, B, T);
      set_bw_weights(n, p, ea);
      p.xa = dG; p.lda = 128; p.cinA = 128;
      p.y = dc; p.ldy = lddc; p.accumulate = (l != L - 1);
      RUN(conv_go(p, MODE_PLAIN, precise, s));
    }
    {  // dX_l = dxo*sqrt(.5) + convT(dG)   (kind 1, l == 0: times LeakyReLU'(X_0))
      float* out = dXall + (long long)l * P;
      ConvP p = base_conv(n, B, T);
      set_bw_weights(n, p, ec);
      p.xa = dG; p.lda = 128; p.cinA = 128;
      p.ktaps = ec.k; p.dil = dil; p.off0 = -off0 - (ec.k - 1) * dil;
      p.y = out; p.ldy = 64;
      // conv input was dropout(x): the conv path goes through the regenerated keep mask
      if (d.dropout > 0.f) { p.epi_drop_p = d.dropout; p.epi_drop_seed = layer_seed(seed_val, l); p.drop_seed_ptr = seed_ptr; }
      if (dxo) { p.res = dxo; p.ldr = 64; p.res_scale = rs; }
      if (l == 0 && d.kind == 1) { p.dmask = X; p.ldm = 64; p.dmask_act = ACT_LRELU; }
      RUN(conv_go(p, MODE_PLAIN, precise, s));
      dxo = out;
    }
  }
  if (fused) {  // first conv: dx through one transposed 1x1; weight gradients of first conv + head from the planes
    if (dx && !bfold) {
      PsP p = ps_base(n, B, T, params);
      p.x = dxo; p.ldx = 64; p.cin = 64; p.y = dx; p.ldy = lddx; p.out_scale = dx_scale;
      p.layers = n->d_ps + 3 * PS_MAXL; p.L = 1;
      RUN(pstack_plan(p, Tb.t[3], precise));
      RUN(launch_pstack(p, precise, ps_flops(Tb.t[3], 1, N), s));
    }
    if (want_w) {
      if (defer_wn && !precise && ws == s) {  // with the weight-norm backward: one launch for all stacks of the model
        n->pw_pending = true; n->pw_B = B; n->pw_T = T; n->pw_a = s16; n->pw_b = f16;
        n->pw_params = plain_wgrad_params(n, B, T, s16, f16);  // tables / slot counts of THIS shape
      } else {
        RUN(plain_wgrad(n, B, T, s16, f16, precise, ws));
      }
    }
  } else
  {  // first conv
    const ConvEntry& e = n->ents[n->idx_first];
    if (want_w) {
      WgradP w = base_wgrad(n, B, T);
      w.a1 = dxo; w.lda1 = 64; w.ca1 = 64; w.ca = 64;
      w.x = x; w.ldx = ldx; w.cx = e.cin;
      wgrad_slots(n, n->idx_first, B, T, w);
      RUN(wgrad_go(n, w, precise));
    }
    if (dx) {
      ConvP p = base_conv(n, B, T);
      set_bw_weights(n, p, e);
      p.xa = dxo; p.lda = 64; p.cinA = 64;
      p.y = dx; p.ldy = lddx; p.out_scale = dx_scale;
      RUN(conv_go(p, MODE_PLAIN, precise, s));
    }
  }
  if (want_w) {
    RUN(wgrad_flush(n, B, T, precise, ws));
    RUN(finish_wnorm(n, params, grads, defer_wn, ws));
    RUN(join_wgrad(n, s, ws));
  }
  return CRK_OK;
}


__global__ void seed_next_kernel(unsigned long long* state, unsigned long long* out) {
  // splitmix64 of a Weyl sequence: distinct, well-mixed seeds; the per-layer / per-element hashing is dropout_scale's
  unsigned long long z = (*state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  *out = (z ^ (z >> 31)) & 0x3fffffffffffffffull;
}
extern "C" int crk_seed_next(unsigned long long* state, unsigned long long* out, void* stream) {
  if (!state || !out) return CRK_ERR_ARG;
  hipLaunchKernelGGL(seed_next_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, out);
  CRK_CHECK_LAUNCH();
  return CRK_OK;
}

// ---- several nets at once (the sub-nets of a model share one optimizer step) ---------------------------------
// The deferred weight-norm backward of every net that has one pending (crk_net_backward with CRK_FLAG_DEFER_WNORM),
// in ONE launch.  Nets without pending work are skipped.
static int flush_plain_wgrads(int n_nets, void* const* nets, hipStream_t s) {
  {  // the deferred weight gradients of the plain convs (first conv + head of every stack), one launch
    PwMP M; memset(&M, 0, sizeof(M));
    int layers = 0, max_G = 0, max_wa = 0, max_wb = 0, max_tiles = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n_nets; i++) {
      Net* n = (Net*)nets[i];
      if (!n) return CRK_ERR_ARG;
      if (!n->pw_pending) continue;
      if (M.n == CRK_MAX_NETS_PW) { RUN(flush_pending_plain_wgrad(n, s)); continue; }
      PsTables Tb;
      ps_build(n, (long long)n->pw_B * n->pw_T, Tb);
      M.q[M.n] = n->pw_params;
      M.first[M.n] = layers;
      layers += Tb.nw;
      if (n->pw_params.G > max_G) max_G = n->pw_params.G;
      if (Tb.max_wa > max_wa) max_wa = Tb.max_wa;
      if (Tb.max_wb > max_wb) max_wb = Tb.max_wb;
      for (int j = 0; j < Tb.nw; j++) {
        const int tiles = ((Tb.w[j].ca + 31) / 32) * ((Tb.w[j].cb + 31) / 32) * Tb.w[j].k;
        if (tiles > max_tiles) max_tiles = tiles;
      }
      flops += Tb.wflops_per_frame * n->pw_B * n->pw_T;
      bytes += 2.0 * (Tb.max_wa + Tb.max_wb) * (double)n->pw_B * n->pw_T * Tb.nw;
      M.n++;
      n->pw_pending = false;
    }
    M.first[M.n] = layers;
    if (M.n > 0) RUN(launch_pstack_wgrad_multi(M, layers, max_G, max_wa, max_wb, max_tiles, flops, bytes, s));
  }
  return CRK_OK;
}
extern "C" int crk_nets_wnorm_bwd(int n_nets, void* const* nets, void* stream) {
  if (n_nets < 0 || (n_nets > 0 && !nets)) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  RUN(flush_plain_wgrads(n_nets, nets, s));
  NetRefs R; memset(&R, 0, sizeof(R));
  int total = 0;
  for (int i = 0; i < n_nets; i++) {
    Net* n = (Net*)nets[i];
    if (!n) return CRK_ERR_ARG;
    if (!n->wn_pending) continue;
    RUN(wait_side_work(n, s));  // (weight gradients on a side stream: their partial sums first)
    if (R.n == CRK_MAX_NETS) {  // more nets than one launch holds: this one goes alone
      RUN(flush_pending_wnorm(n, s));
      continue;
    }
    NetRef& q = R.r[R.n++];
    q.ents = n->wn_ents ? n->wn_ents : n->d_ents; q.n_ents = (int)n->ents.size(); q.first = total;
    q.params = n->wn_params; q.grads = n->wn_grads; q.partials = n->partials; q.norms = n->norms;
    total += q.n_ents;
    n->wn_pending = false;
  }
  if (R.n == 0) return CRK_OK;
  return launch_wnorm_bwd_multi(R, total, s);
}

// Weight preparation (weight-norm fold + bf16 operand planes) of every net whose parameters changed, in ONE launch;
// what crk_net_forward / crk_net_backward would do one net at a time on their first call after an optimizer step.
// params[i]: the parameter block of nets[i]; version: as for crk_net_forward.
extern "C" int crk_nets_prepare(int n_nets, void* const* nets, const float* const* params, unsigned long long version,
                                float* bump_step, void* stream) {
  if (n_nets < 0 || (n_nets > 0 && (!nets || !params))) return CRK_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  NetRefs R; memset(&R, 0, sizeof(R));
  int total = 0, nmax = 1;
  for (int i = 0; i < n_nets; i++) {
    Net* n = (Net*)nets[i];
    if (!n || !params[i]) return CRK_ERR_ARG;
    if (n->prepared_version == version && n->prepared_params == params[i]) continue;
    if (net_nmax(n) > nmax) nmax = net_nmax(n);
    if (R.n == CRK_MAX_NETS) { RUN(ensure_prepared(n, params[i], version, s)); continue; }
    if (n->Gs == 0) RUN(upload_entries(n, 1, 1));
    RUN(wait_side_work(n, s));
    NetRef& q = R.r[R.n++];
    q.ents = n->d_ents; q.n_ents = (int)n->ents.size(); q.first = total;
    q.params = params[i]; q.whi = n->whi; q.wlo = n->wlo; q.norms = n->norms;
    total += q.n_ents;
    n->prepared_version = version; n->prepared_params = params[i];
  }
  if (R.n == 0) return bump_step ? launch_step_bump(bump_step, s) : CRK_OK;
  R.bump = bump_step;
  return launch_weight_prep_multi(R, total, nmax, s);
}
