// `make experiments`: compile check of the rejected kernels against the current product headers (device code only; nothing links
// or launches them).
#include "dsg_rejected_kernels.h"
#include "dsg_stream_ln.h"
namespace dsg {
template __global__ void k_gemm_tp<PBF16, PRO_LN, EPI_GELU, 256, 128>(const GemmArgs);
template __global__ void k_gemm_tp<PBF16, PRO_DIRECT, EPI_RESID, 256, 128>(const GemmArgs);
template __global__ void k_qkv_attn<PBF16, 64, 6, 256>(const QkvAttnArgs);
template __global__ void k_ws_ln<EPI_QKV, 16>(const GemmArgs);
template __global__ void k_ws_ln<EPI_OUT, 16>(const GemmArgs);
template __global__ void k_attn_op2<PBF16, 4, 6>(const AttnOpArgs);
}  // namespace dsg
