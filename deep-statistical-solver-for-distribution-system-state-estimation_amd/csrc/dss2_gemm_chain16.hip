// bf16x6 instantiations of the layer-chain kernel (dss2_gemm_chain_kernel.hpp, B16 = true).
//
// A translation unit of its own because it is compiled WITHOUT packed fp32 VALU ops (build.sh: -packed-fp32-ops for this
// file, dss2_wgrad16.hip and dss2_stack.hip; the build disassembles them and fails if one v_pk_*_f32 is left).
// Root cause (round 3, profiles/r03_pk_fma_investigation.txt, tools/micro/pkfma_beside_mfma.hip): on MI355X a v_pk_fma_f32
// / v_pk_add_f32 / v_pk_mul_f32 of one wave occasionally returns wrong values in lanes 48..63 -- the last 16-lane pass of the
// op -- while ANOTHER wave of the same SIMD streams v_mfma_f32_32x32x16_bf16, with two or more workgroups per CU.  An 87-line
// kernel without LDS whose failing wave executes no MFMA at all reproduces it (fp32 MFMAs or plain VALU work beside it: never;
// one workgroup per CU: never), so it is neither an LDS race of this kernel nor a missed MFMA -> VALU hazard of the compiler.
// Round 2 met it as run-to-run different values of the epilogue's `v += prebias[m] * rowscale[row][m]`; with today's paired
// operand split (split3_pair -> v_pk_add_f32) a packed build returns garbage in every lane (tools/pk_stress.py).  Plain VALU
// ops are also the cheaper fillers beside MFMAs (MI355X_MICROARCH.md).
#include <stdlib.h>

#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

int launch_chain16(const dss2_gemm_prop_args& a, const ChainTable& ct, int rsplit, hipStream_t s) {
#define DSS2_CASE16(NRB, NMAT)                                                           \
  if (a.nrb == NRB && a.nmat == NMAT && rsplit == 1)                                     \
    return a.ncg <= 4 ? launch_chain<NRB, NMAT, 4, 1, true>(a, ct, s) : launch_chain<NRB, NMAT, 8, 1, true>(a, ct, s);
#define DSS2_CASE16_2(NRB, NMAT)                                                         \
  if (a.nrb == NRB && a.nmat == NMAT && rsplit == 2)                                     \
    return 2 * a.ncg <= 4 ? launch_chain<NRB, NMAT, 4, 2, true>(a, ct, s) : launch_chain<NRB, NMAT, 8, 2, true>(a, ct, s);
  DSS2_CASE16(1, 2) DSS2_CASE16(1, 3) DSS2_CASE16(1, 4) DSS2_CASE16(2, 2) DSS2_CASE16(2, 3)
  DSS2_CASE16(3, 2) DSS2_CASE16(4, 2)
  // 96-row tiles: three waves per column group (one row block each, twelve waves = three per SIMD) instead of one wave per SIMD
  // with three row blocks -- the VALU-issue-bound phases of a wave run beside the other waves' MFMAs (DSS2_CHAIN_RS3=0: four waves)
  static const int rs3 = [] { const char* e = getenv("DSS2_CHAIN_RS3"); return e ? atoi(e) : 1; }();
  if (a.nrb == 3 && a.nmat == 3 && rsplit == 1 && a.ncg <= 4 && rs3) return launch_chain<3, 3, 12, 3, true>(a, ct, s);
  if (a.nrb == 3 && a.nmat == 3 && rsplit == 1 && a.ncg <= 4) return launch_chain<3, 3, 4, 1, true>(a, ct, s);   // (8 waves would spill)
  DSS2_CASE16_2(2, 2) DSS2_CASE16_2(2, 3) DSS2_CASE16_2(2, 4) DSS2_CASE16_2(4, 2)
#undef DSS2_CASE16
#undef DSS2_CASE16_2
  set_error("gemm_prop_chain(bf16x6): unsupported (nrb=%d, nmat=%d, row split %d)", a.nrb, a.nmat, rsplit);
  return 2;
}

}  // namespace dss2

#ifdef DSS2_CHAIN_STAMPS
extern "C" int dss2_debug_read_cstamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dss2::g_cstamps), sizeof(unsigned long long) * n);
}
#endif
