// Stand-alone launches of the one-wavefront-per-(window quad, head) T = 16 kernels (bodies: attn16_wave.h); the layer executor's
// default path issues them merged with the other occupancy levels (attention_coop.hip).
#include "attn16_wave.h"
using namespace t16w;
namespace {
template <int DH>
__global__ __launch_bounds__(256) void k_attn_t16_fwd(A16Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t16k[];
  t16_fwd_body<DH>(A, blockIdx.x, smem_t16k);
}
template <int DH>
__global__ __launch_bounds__(256) void k_attn_t16_bwd(A16BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t16k[];
  t16_bwd_body<DH>(A, blockIdx.x, gridDim.x, smem_t16k);
}
}  // namespace

// bf16 I/O, T = 16, H % 4 == 0; called from attention.hip's entry points
int gd_attn_t16_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len, int n_win,
                    int d, int H, const float* tau, float tau_min, hipStream_t st) {
  A16Args A{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const dim3 grid((unsigned)(gd_div_up(n_win, kWinPerWave) * (H / 4)));
  if (d / H == 16) hipLaunchKernelGGL((k_attn_t16_fwd<16>), grid, dim3(256), 4 * kWaveLds, st, A);
  else hipLaunchKernelGGL((k_attn_t16_fwd<32>), grid, dim3(256), 4 * kWaveLds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

int gd_attn_t16_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                    const int* win_start, const int* win_len, int n_win, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  A16BwdArgs A{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)dout, (unsigned short*)dqk, (unsigned short*)dv,
               dtau_part, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const dim3 grid((unsigned)(gd_div_up(n_win, kWinPerWave) * (H / 4)));
  if (d / H == 16) hipLaunchKernelGGL((k_attn_t16_bwd<16>), grid, dim3(256), 4 * kWaveLds, st, A);
  else hipLaunchKernelGGL((k_attn_t16_bwd<32>), grid, dim3(256), 4 * kWaveLds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}
