// bf16x6 instantiations of the layer-chain kernel (dss2_gemm_chain_kernel.hpp, B16 = true).
//
// A translation unit of its own because it is compiled WITHOUT packed fp32 VALU ops (build.sh: -packed-fp32-ops for this
// file only).  With them the compiler turns the epilogue's `v += prebias[m] * rowscale[row][m]` into v_pk_fma_f32 with
// op_sel broadcasts, and those returned wrong values in lanes 48..63 -- run-to-run different -- whenever a second
// workgroup's v_mfma_f32_32x32x16_bf16 stream shared the SIMD (two workgroups per CU, > 256 tiles); the same source with
// v_fmac_f32 (or one workgroup per CU, or the fp32 MFMA instantiation) is bitwise reproducible and matches the fp32 path
// to 5e-7 (tools/accuracy_bf16x6.py; DESIGN.md section 4.2 has the bisection).  Plain VALU ops are also the cheaper
// fillers beside MFMAs (MI355X_MICROARCH.md).
#include "dss2_gemm_chain_kernel.hpp"

namespace dss2 {

int launch_chain16(const dss2_gemm_prop_args& a, const ChainTable& ct, int rsplit, hipStream_t s) {
#define DSS2_CASE16(NMAT)                                                                                      \
  if (a.nmat == NMAT) return rsplit == 2 ? launch_chain<2, NMAT, 4, 2, true>(a, ct, s) : launch_chain<2, NMAT, 4, 1, true>(a, ct, s);
  DSS2_CASE16(2) DSS2_CASE16(3)
#undef DSS2_CASE16
  if (a.nmat == 4 && rsplit == 2) return launch_chain<2, 4, 4, 2, true>(a, ct, s);
  set_error("gemm_prop_chain(bf16x6): unsupported (nmat=%d, row split %d)", a.nmat, rsplit);
  return 2;
}

}  // namespace dss2
