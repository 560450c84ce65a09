// Layer-chained variant of gemm_prop: host side and the fp32 instantiations (kernel: dss2_gemm_chain_kernel.hpp).
#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

// row split: two waves per column group for narrow layers on two-row-block tiles (DSS2_CHAIN_RS overrides: 1 or 2)
static int chain_row_split(int nrb, int ncg) {
  static const int forced = [] { const char* e = getenv("DSS2_CHAIN_RS"); return e ? atoi(e) : 0; }();
  if (nrb % 2 != 0 || 2 * ncg > 8) return 1;
  if (forced == 1 || forced == 2) return forced;
  return ncg <= 2 ? 2 : 1;
}

}  // namespace dss2

extern "C" int dss2_gemm_prop_chain_supported(int nrb, int nmat, int kreal, int hout, int ell_width) {
  using namespace dss2;
  if (nmat < 2 || nmat > 4 || nrb * nmat > 9 || !(nrb == 1 || nrb == 2 || nrb == 3 || nrb == 4)) return 0;
  if (kreal != hout || (hout & 3) != 0 || hout > 256 || ell_width <= 0 || ell_width > 32) return 0;
  const int kpad = (kreal + 7) / 8 * 8, ncg = (hout + 31) / 32;
  return chain_lds_bytes(nrb, kpad, ncg, ell_width) <= (size_t)kMaxLdsBytes ? 1 : 0;
}

extern "C" int dss2_gemm_prop_chain_gate_words(int nrb, int nmat, int kreal, int hout, int ell_width) {
  using namespace dss2;
  dss2_gemm_prop_args a = {};
  a.b_format = 1; a.nrb = nrb; a.nmat = nmat; a.kreal = kreal; a.kpad = (kreal + 15) / 16 * 16; a.hout = hout; a.ncg = (hout + 31) / 32; a.ell_width = ell_width;
  if (!(kreal == hout && (hout & 3) == 0 && ell_width > 0)) return 0;
  const bool tall = (nrb == 6 || nrb == 3) && chain_sp6_supported(a);
  const bool sp64 = nrb == 2 && chain_row_split(nrb, a.ncg) == 1 && chain_sp_supported(a);      // (round 4: the 64-row split-plane chain too)
  if (!tall && !sp64) return 0;
  return a.ncg * 32 * ((4 * nrb + 7) / 8);      // per wave (column group): 64 lanes x ceil(row pieces / 8) 32-bit words (dss2_gemm_chain_sp6.hip, _sp.hip)
}

extern "C" int dss2_gemm_prop_chain16_supported(int nrb, int nmat, int kreal, int hout, int ell_width) {
  using namespace dss2;
  if (nrb == 6) {      // 192-row tiles: only the split-plane form exists (dss2_gemm_chain_sp6.hip)
    dss2_gemm_prop_args a = {};
    a.b_format = 1; a.nrb = nrb; a.nmat = nmat; a.kreal = kreal; a.kpad = (kreal + 15) / 16 * 16; a.hout = hout; a.ncg = (hout + 31) / 32; a.ell_width = ell_width;
    return (kreal == hout && (hout & 3) == 0 && ell_width > 0 && chain_sp6_supported(a)) ? 1 : 0;
  }
  // the shapes of the fp32 chain whose bf16x6 instantiation exists without register spills (dss2_gemm_chain16.hip)
  if (!dss2_gemm_prop_chain_supported(nrb, nmat, kreal, hout, ell_width)) return 0;
  const int ncg = (hout + 31) / 32, rsplit = chain_row_split(nrb, ncg);
  if (nrb == 2 && nmat == 4 && rsplit != 2) return 0;
  if (nrb == 3 && nmat == 3 && ncg > 4) return 0;
  return chain_lds_bytes(nrb, (kreal + 15) / 16 * 16, ncg, ell_width, chain_rm(nrb, rsplit, true, nmat) ? nmat : 0) <= (size_t)kMaxLdsBytes ? 1 : 0;
}

static int chain_impl(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, const dss2_chain_head* head, void* stream);

// b_format 2 (weights as two fp16 planes + scale exponents, tile GEMM as f16x3): the split-plane chain of 64-row tiles on 16x16x32 MFMAs
extern "C" int dss2_gemm_prop_chain_f16_supported(int nrb, int nmat, int kreal, int hout, int ell_width) {
  using namespace dss2;
  if (kreal != hout || ell_width <= 0 || (hout & 3) != 0) return 0;
  dss2_gemm_prop_args a = {};
  a.b_format = 2; a.nrb = nrb; a.nmat = nmat; a.kreal = kreal; a.kpad = (kreal + 15) / 16 * 16; a.hout = hout; a.ncg = (hout + 31) / 32; a.ell_width = ell_width;
  if (nrb == 3 || nrb == 6) return chain_sp6_supported(a) ? 1 : 0;      // 96- / 192-row tiles (dss2_gemm_chain_sp6.hip)
  if (nrb != 2 || (hout & 31) != 0 || !dss2_gemm_prop_chain16_supported(nrb, nmat, kreal, hout, ell_width)) return 0;
  return chain_row_split(nrb, a.ncg) == 1 && chain_sp_supported(a) ? 1 : 0;
}

static int dss2_gemm_prop_chain_launch(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, void* stream);
extern "C" int dss2_gemm_prop_chain(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, void* stream) {
  if (!ap) { dss2::set_error("dss2_gemm_prop_chain: null argument"); return 2; }
  DSS2_RECORD([a = *ap, l = dss2::plan_keep(layers, (size_t)(n_layers > 0 ? n_layers : 0)), n_layers](void* s_) { return dss2_gemm_prop_chain_launch(&a, dss2::plan_ptr(l), n_layers, s_); });
  return dss2_gemm_prop_chain_launch(ap, layers, n_layers, stream);
}
static int dss2_gemm_prop_chain_launch(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, void* stream) {
  return chain_impl(ap, layers, n_layers, nullptr, stream);
}

extern "C" int dss2_gemm_prop_chain_head_supported(int nrb, int nmat, int kreal, int hout, int ell_width, int nout) {
  using namespace dss2;
  if (nout < 1 || nout > 4 || !dss2_gemm_prop_chain16_supported(nrb, nmat, kreal, hout, ell_width)) return 0;
  dss2_gemm_prop_args a = {};
  a.b_format = 1; a.nrb = nrb; a.nmat = nmat; a.kreal = kreal; a.kpad = (kreal + 15) / 16 * 16; a.hout = hout; a.ncg = (hout + 31) / 32; a.ell_width = ell_width;
  if (chain_row_split(nrb, a.ncg) == 1 && chain_sp_supported(a)) return 3;      // 64-row tiles: forward (bit 0) and backward (bit 1) head
  if ((nrb == 3 || nrb == 6) && chain_sp6_supported(a)) return 2;               // 96- / 192-row tiles: the backward head only
  return 0;
}

extern "C" int dss2_gemm_prop_chain_head_wgrad_supported(int nrb, int nmat, int kreal, int hout, int ell_width, int nout) {
  using namespace dss2;
  if (nout < 1 || nout > 2 || !(dss2_gemm_prop_chain_head_supported(nrb, nmat, kreal, hout, ell_width, nout) & 2)) return 0;
  if (nrb == 3 || nrb == 6) return 1;      // the split-plane chains of 96- / 192-row tiles (dss2_gemm_chain_sp6.hip)
  dss2_gemm_prop_args a = {};
  a.b_format = 1; a.nrb = nrb; a.nmat = nmat; a.kreal = kreal; a.kpad = (kreal + 15) / 16 * 16; a.hout = hout; a.ncg = (hout + 31) / 32; a.ell_width = ell_width;
  return nrb == 2 && chain_row_split(nrb, a.ncg) == 1 && chain_sp_supported(a) ? 1 : 0;      // the 64-row split-plane chain (dss2_gemm_chain_sp.hip)
}

static int dss2_gemm_prop_chain_head_launch(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, const dss2_chain_head* head, void* stream);
extern "C" int dss2_gemm_prop_chain_head(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, const dss2_chain_head* head, void* stream) {
  if (!ap) { dss2::set_error("dss2_gemm_prop_chain_head: null argument"); return 2; }
  if (!head) { dss2::set_error("dss2_gemm_prop_chain_head: head is NULL"); return 2; }
  DSS2_RECORD([a = *ap, l = dss2::plan_keep(layers, (size_t)(n_layers > 0 ? n_layers : 0)), n_layers, h = *head](void* s_) { return dss2_gemm_prop_chain_head_launch(&a, dss2::plan_ptr(l), n_layers, &h, s_); });
  return dss2_gemm_prop_chain_head_launch(ap, layers, n_layers, head, stream);
}
static int dss2_gemm_prop_chain_head_launch(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, const dss2_chain_head* head, void* stream) {
  using namespace dss2;
  if (!head || (head->mode != 1 && head->mode != 2)) { set_error("gemm_prop_chain_head: head.mode must be 1 or 2"); return 2; }
  if (!(dss2_gemm_prop_chain_head_supported(ap->nrb, ap->nmat, ap->kreal, ap->hout, ap->ell_width, head->nout) & head->mode) || (ap->b_format != 1 && ap->b_format != 2) ||
      (ap->b_format == 2 && !dss2_gemm_prop_chain_f16_supported(ap->nrb, ap->nmat, ap->kreal, ap->hout, ap->ell_width))) {
    set_error("gemm_prop_chain_head: unsupported shape (nrb=%d nmat=%d hid=%d nout=%d b_format=%d)", ap->nrb, ap->nmat, ap->hout, head->nout, ap->b_format);
    return 2;
  }
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  for (int m = 0; m < ap->nmat; ++m)
    if (!head->W[m] || !al16(head->W[m])) { set_error("gemm_prop_chain_head: W[%d] missing or misaligned", m); return 2; }
  if (head->mode == 1 && !head->Y) { set_error("gemm_prop_chain_head: forward head needs Y"); return 2; }
  if (head->mode == 2 && (!head->G || !head->Xout || !al16(head->Xout) || !al16(head->gate) || (head->ldxo & 3) || (head->ld_gate & 3))) {
    set_error("gemm_prop_chain_head: backward head needs G, a 16-byte aligned Xout (and gate)"); return 2;
  }
  if (head->drop_id && !ap->drop_state) { set_error("gemm_prop_chain_head: drop_id without drop_state"); return 2; }
  if (head->wg_slab && (head->mode != 2 || !head->gate || !dss2_gemm_prop_chain_head_wgrad_supported(ap->nrb, ap->nmat, ap->kreal, ap->hout, ap->ell_width, head->nout))) {
    set_error("gemm_prop_chain_head: wg_slab needs mode 2, gate and a shape dss2_gemm_prop_chain_head_wgrad_supported accepts"); return 2;
  }
  return chain_impl(ap, layers, n_layers, head, stream);
}

static int chain_impl(const dss2_gemm_prop_args* ap, const dss2_chain_layer* layers, int n_layers, const dss2_chain_head* head, void* stream) {
  using namespace dss2;
  const dss2_gemm_prop_args& a = *ap;
  if (n_layers < 1 || n_layers > CHAIN_MAX || !layers) { set_error("gemm_prop_chain: 1..%d layers, got %d", CHAIN_MAX, n_layers); return 2; }
  if (a.ntiles <= 0) return 0;
  const bool tall16 = (a.b_format == 1 || a.b_format == 2) && a.nrb == 6 && dss2_gemm_prop_chain16_supported(a.nrb, a.nmat, a.kreal, a.hout, a.ell_width);
  if ((!tall16 && !dss2_gemm_prop_chain_supported(a.nrb, a.nmat, a.kreal, a.hout, a.ell_width)) || !a.ell_tiles || a.prop_in || a.narrow_h ||
      a.rowscale || a.kpad != (a.b_format >= 1 ? (a.kreal + 15) / 16 * 16 : (a.kreal + 7) / 8 * 8) || a.ncg != (a.hout + 31) / 32) {
    set_error("gemm_prop_chain: unsupported shape (nrb=%d nmat=%d k=%d hout=%d ell=%d); use dss2_gemm_prop per layer",
              a.nrb, a.nmat, a.kreal, a.hout, a.ell_width);
    return 2;
  }
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (((!a.X || !al16(a.X)) && !(head && head->mode == 2)) || (a.ldx & 3) || (a.ldy & 3) || (a.ld_relu & 3) || (a.ld_dmask & 3) || (a.ld_add & 3)) {
    set_error("gemm_prop_chain: operands must be 16-byte aligned with leading dimensions divisible by 4"); return 2;
  }
  ChainTable ct = {};
  ct.n = n_layers;
  bool any_pre = false;
  for (int i = 0; i < n_layers; ++i) {
    const dss2_chain_layer& L = layers[i];
    if (!L.Bp || !L.Y || !al16(L.Y) || !al16(L.bias) || !al16(L.relu_src) || !al16(L.dmask) || !al16(L.add_src) || !al16(L.prebias)) {
      set_error("gemm_prop_chain: layer %d has a missing or misaligned operand", i); return 2;
    }
    any_pre = any_pre || L.prebias;
    if (L.drop_id && !a.drop_state) { set_error("gemm_prop_chain: layer %d asks for in-kernel dropout without drop_state", i); return 2; }
    ct.l[i] = L;
  }
  if (any_pre && !a.pre_rowscale) { set_error("gemm_prop_chain: prebias needs pre_rowscale"); return 2; }
  hipStream_t s = as_stream(stream);
  const int rsplit = chain_row_split(a.nrb, a.ncg);
  if (a.b_format == 2) {     // weights packed as two fp16 planes with scale exponents, tile GEMM as f16x3
    if (!dss2_gemm_prop_chain_f16_supported(a.nrb, a.nmat, a.kreal, a.hout, a.ell_width)) {
      set_error("gemm_prop_chain(f16x3): unsupported shape (nrb=%d nmat=%d k=%d hout=%d)", a.nrb, a.nmat, a.kreal, a.hout);
      return 2;
    }
    if (a.nrb == 3 || a.nrb == 6) return launch_chain_sp6(a, ct, head, s);
    return launch_chain_sp(a, ct, head, s);
  }
  if (a.b_format == 1) {     // weights packed as bf16x3 fragments, tile GEMM as bf16x6
    if (!dss2_gemm_prop_chain16_supported(a.nrb, a.nmat, a.kreal, a.hout, a.ell_width) || (a.kpad & 15)) {
      set_error("gemm_prop_chain(bf16x6): unsupported shape (nrb=%d nmat=%d k=%d kpad=%d hout=%d)", a.nrb, a.nmat, a.kreal, a.kpad, a.hout);
      return 2;
    }
    if (rsplit == 1 && chain_sp_supported(a)) return launch_chain_sp(a, ct, head, s);      // 64-row tiles, H >= 96: split-plane form
    if ((a.nrb == 6 || a.nrb == 3) && chain_sp6_supported(a)) return launch_chain_sp6(a, ct, head, s);      // 96- / 192-row tiles: split-plane form, NRB row blocks per wave
    if (head) { set_error("gemm_prop_chain_head: the split-plane chain does not cover this shape"); return 2; }
    return launch_chain16(a, ct, rsplit, s);
  }
  if (a.b_format != 0) { set_error("gemm_prop_chain: unknown b_format %d", a.b_format); return 2; }
#define DSS2_CASE(NRB, NMAT)                                                             \
  if (a.nrb == NRB && a.nmat == NMAT && rsplit == 1)                                     \
    return a.ncg <= 4 ? launch_chain<NRB, NMAT, 4, 1>(a, ct, s) : launch_chain<NRB, NMAT, 8, 1>(a, ct, s);
#define DSS2_CASE2(NRB, NMAT)                                                            \
  if (a.nrb == NRB && a.nmat == NMAT && rsplit == 2)                                     \
    return 2 * a.ncg <= 4 ? launch_chain<NRB, NMAT, 4, 2>(a, ct, s) : launch_chain<NRB, NMAT, 8, 2>(a, ct, s);
  DSS2_CASE(1, 2) DSS2_CASE(1, 3) DSS2_CASE(1, 4) DSS2_CASE(2, 2) DSS2_CASE(2, 3) DSS2_CASE(2, 4)
  DSS2_CASE(3, 2) DSS2_CASE(3, 3) DSS2_CASE(4, 2)
  DSS2_CASE2(2, 2) DSS2_CASE2(2, 3) DSS2_CASE2(2, 4) DSS2_CASE2(4, 2)
#undef DSS2_CASE
#undef DSS2_CASE2
  set_error("gemm_prop_chain: unsupported (nrb=%d, nmat=%d)", a.nrb, a.nmat);
  return 2;
}
