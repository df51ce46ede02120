// attention_kernel.hip -- fused single-head spatial self-attention core for CDNA4 (gfx950), head dimension 512:
//   O = softmax(Q K^T * scale) V     per frame (batch item), Q, K, V = the q / k / v projections of the attention blocks
// Replaces, in the reference: F.scaled_dot_product_attention inside diffusers' AttnProcessor2_0 (AttentionWithExtraDim,
// models/vae_blocks3d_sd3.py:119-147) and xformers.ops.memory_efficient_attention (models/vae_models.py:518-520, 581-583) --
// and, in this library, the five launches it was built from in rounds 1-2 (pack K, QK^T with fp32 scores through HBM, row
// softmax, pack V^T, PV).
//
// One workgroup = 4 wave64 = 128 queries of one frame; every wave owns 32 queries over the FULL head dimension and is alone on
// its SIMD (a 512-register kernel: 256 accumulators of O^T in AGPRs, the wave's Q fragments -- 128 registers -- resident):
//   * S^T = K Q^T per block of 32 keys: 32 x v_mfma_f32_32x32x16 (A = K fragment from LDS, B = Q fragment in registers).  The
//     accumulator lane is the QUERY, its 16 registers are keys -- row i of the K operand carries key pi(i), pi chosen so that the
//     16 keys a half-wave holds are CONTIGUOUS (half h: keys 16h .. 16h+15 of the block, register r <-> key 16h + r);
//   * two passes over the keys: the row maxima first (S only), then P = exp2(S - max) and the second product -- no accumulator is
//     ever rescaled (an online softmax was built first: the rescale of 256 accumulators per lane spilled ~280 registers);
//   * the probabilities never leave their lanes: registers 8s .. 8s+7 of a half-wave, rounded to the storage dtype, ARE the B
//     operand (k = keys 8s.. and 16+8s.. of the block) of the second product O^T += V^T P^T -- 32 x MFMA per key block (16
//     d-blocks x 2 key steps), A = V^T fragment from LDS ([d][key]: contiguous keys, from the pre-transposed V).
// K and V^T blocks are double-buffered in LDS, filled by global -> LDS DMA (K row pitch 1040 bytes, V^T chunk-swizzled:
// conflict-free ds_read_b128).  fp32 accumulation throughout; exp2 with the scale folded in.  Keys beyond N are masked; queries beyond N are
// computed on clamped rows and not stored.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cvvae.h"
#include "conv_kernel.h"

namespace cvvae {

constexpr int FA_D = 512, FA_KB = 32, FA_QW = 32, FA_NW = 4;
constexpr int FA_KPITCH = FA_D * 2 + 16;    // bytes per key row in LDS
constexpr int FA_KBYTES = FA_KB * FA_KPITCH, FA_VBYTES = FA_D * FA_KB * 2;  // V^T block: 512 rows x 64 bytes, chunk-swizzled
constexpr int FA_BUF = FA_KBYTES + FA_VBYTES;

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void flash_attn_d512_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vt, T* __restrict__ o, int N, long long ldvt, float scale_log2) {
  using v8 = typename Tr<T>::v8;
  __shared__ __attribute__((aligned(16))) char smem[2 * FA_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const long long bt = blockIdx.y;
  const int q0 = blockIdx.x * (FA_QW * FA_NW) + wave * FA_QW;
  const int qi = q0 + l31 < N ? q0 + l31 : N - 1;  // (clamped: rows beyond N are computed and dropped)
  const T* qrow = q + (bt * N + qi) * FA_D + hf * 8;
  v8 qf[FA_D / 16];
#pragma unroll
  for (int t = 0; t < FA_D / 16; ++t) qf[t] = *reinterpret_cast<const v8*>(qrow + t * 16);

  const T* kb_base = k + bt * (long long)N * FA_D;
  const T* vt_base = vt + bt * (long long)FA_D * ldvt;
  const int nkb = (N + FA_KB - 1) / FA_KB;
  // Staging: global -> LDS DMA (global_load_lds_dwordx4: lane i of a wave-load writes 16 bytes at dst + 16 i, no VGPRs involved), so
  // the NEXT block's K and V^T are requested at the top of an iteration and have the whole iteration to arrive -- with one wave
  // per SIMD nothing else hides that latency (a register-staged version, loads before a product and LDS writes after it, ran at
  // 120 TFLOP/s).  A wave-load is 1 KiB: one key row of K (LDS row pitch 1040 bytes), or 16 rows x 64 bytes of V^T, whose four
  // 16-byte chunks per row are XOR-swizzled by (row >> 2) & 3 so that the fragment reads below stay bank-conflict-free without
  // padding.  Every wave issues 8 + 8 wave-loads per block and waits for its own (s_waitcnt vmcnt(0)) before the barrier.
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto dma_k = [&](int kb, int buf) {
    char* kd = smem + buf * FA_BUF;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = wave * 8 + i;
      int kg = kb * FA_KB + key;
      kg = kg < N ? kg : N - 1;  // (partial last block: clamped rows, masked scores)
      __builtin_amdgcn_global_load_lds((gptr_t)(kb_base + (long long)kg * FA_D + lane * 8), (lptr_t)(kd + key * FA_KPITCH), 16, 0, 0);
    }
  };
  // my chunk of a V^T wave-load: position `lane` holds row 16g + (lane >> 2), logical 16-byte chunk (lane & 3) ^ ((row >> 2) & 3)
  const int v_r = lane >> 2;
  auto dma_v = [&](int kb, int buf) {
    char* vd = smem + buf * FA_BUF + FA_KBYTES;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = wave * 8 + j;
      const int d = 16 * g + v_r;
      const int c = (lane & 3) ^ ((d >> 2) & 3);
      __builtin_amdgcn_global_load_lds((gptr_t)(vt_base + (long long)d * ldvt + (long long)kb * FA_KB + c * 8), (lptr_t)(vd + g * 1024), 16, 0, 0);
    }
  };
  auto dma_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // row i of the K operand carries key pi(i): accumulator register r of half h <-> MFMA row (r&3) + 8(r>>2) + 4h, and we want that
  // row to be key 16h + r of the block, so the lane that FEEDS row i (lane l31 = i of the A operand) reads key
  //   pi(i) = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
  const int pi_key = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
  const unsigned k_off = (unsigned)(pi_key * FA_KPITCH + hf * 16);
  // V^T fragment of d-block b (rows 32b + l31), key step st: logical chunk c = 2 hf + st, stored at chunk c ^ ((row >> 2) & 3) of the row
  const unsigned v_row = (unsigned)((l31 >> 4) * 1024 + (l31 & 15) * 64);   // (+ b * 2048: a d-block is two 16-row groups)
  const unsigned v_sw = (unsigned)((l31 >> 2) & 3);                         // ((32b + l31) >> 2) & 3 = (l31 >> 2) & 3
  // scores of one key block, in log2 units, masked: register r of half hf <-> key kb*32 + 16*hf + r
  auto scores = [&](const char* kd, int kb) -> f32x16 {
    // 32 k16 steps over the head dimension; the K fragments run through a ring of RING registers sets (read RING-1 steps ahead: with
    // one wave per SIMD nothing else covers the LDS latency -- the compiler's own order was read -> wait -> MFMA, 4x slower), and
    // two accumulators break the dependent-MFMA chain
    constexpr int RING = 4, NT = FA_D / 16;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
    v8 kf[RING];
#pragma unroll
    for (int t = 0; t < RING - 1; ++t) kf[t] = *reinterpret_cast<const v8*>(kd + k_off + t * 32);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + RING - 1 < NT) kf[(t + RING - 1) % RING] = *reinterpret_cast<const v8*>(kd + k_off + (t + RING - 1) * 32);
      if (t & 1) s1 = Tr<T>::mfma(kf[t % RING], qf[t], s1);
      else s0 = Tr<T>::mfma(kf[t % RING], qf[t], s0);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 sc;
    const int key0 = kb * FA_KB + 16 * hf;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = (key0 + r < N) ? (s0[r] + s1[r]) * scale_log2 : -1.0e30f;
    return sc;
  };

  // ---- pass 1: the row maxima (S only; K blocks double-buffered).  Two passes instead of an online softmax: rescaling 256
  //      accumulators per lane whenever a maximum moves costs more registers and VALU work than recomputing S (32 of 96 MFMAs per
  //      key block), and the final maximum makes every exponent <= 0 without any correction term.
  float m_run = -1.0e30f;
  // (pass 1 multiplies only 32 MFMAs per block -- shorter than a memory round trip -- so its K blocks run TWO ahead through three
  //  K-sized buffers carved out of pass 2's LDS)
  constexpr int P1B = 3;
  static_assert(P1B * FA_KBYTES <= 2 * FA_BUF, "pass-1 K ring must fit the pass-2 buffers");
  auto dma_k1 = [&](int kb) {
    char* kd = smem + (kb % P1B) * FA_KBYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = wave * 8 + i;
      int kg = kb * FA_KB + key;
      kg = kg < N ? kg : N - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(kb_base + (long long)kg * FA_D + lane * 8), (lptr_t)(kd + key * FA_KPITCH), 16, 0, 0);
    }
  };
  for (int i = 0; i < P1B - 1 && i < nkb; ++i) dma_k1(i);
  for (int kb = 0; kb < nkb; ++kb) {
    // block kb must have landed: only the one younger block's 8 loads may still be in flight
    if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (everyone's part of block kb is in LDS, and everyone has finished block kb - 1, whose buffer is refilled next)
    if (kb + P1B - 1 < nkb) dma_k1(kb + P1B - 1);
    const f32x16 sc = scores(smem + (kb % P1B) * FA_KBYTES, kb);
#pragma unroll
    for (int r = 0; r < 16; ++r) m_run = sc[r] > m_run ? sc[r] : m_run;
  }
  __syncthreads();
  {
    const float mo = __shfl_xor(m_run, 32, 64);  // the partner lane holds the other 16 keys of every block of my query
    m_run = mo > m_run ? mo : m_run;
  }

  // ---- pass 2: P = exp2(S - max), l = sum P, O^T += V^T P^T
  f32x16 oacc[FA_D / 32];
#pragma unroll
  for (int b = 0; b < FA_D / 32; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[b][r] = 0.f;
  float l_run = 0.f;  // my half's partial sum
  dma_k(0, 0);
  dma_v(0, 0);
  dma_wait();
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) {
      dma_k(kb + 1, cur ^ 1);
      dma_v(kb + 1, cur ^ 1);
    }
    const char* kd = smem + cur * FA_BUF;
    const char* vd = kd + FA_KBYTES;
    const f32x16 sc = scores(kd, kb);
    v8 pf[2];  // the probabilities, rounded to the storage dtype: the B operand of the second product
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(sc[r] - m_run);
      pf[r >> 3][r & 7] = (T)pr;
      l_run += pr;
    }
    // (k slot (step st, half, j) <-> key 16*half + 8*st + j of the block = my registers 8*st + j); V^T fragments through a ring too
    {
      constexpr int RING = 4, NV = 2 * (FA_D / 32);  // fragment index f = 2 b + st
      auto vaddr = [&](int f) { return vd + (unsigned)((f >> 1) * 2048) + v_row + (((unsigned)(2 * hf + (f & 1)) ^ v_sw) << 4); };
      v8 vf[RING];
#pragma unroll
      for (int f = 0; f < RING - 1; ++f) vf[f] = *reinterpret_cast<const v8*>(vaddr(f));
#pragma unroll
      for (int f = 0; f < NV; ++f) {
        if (f + RING - 1 < NV) vf[(f + RING - 1) % RING] = *reinterpret_cast<const v8*>(vaddr(f + RING - 1));
        oacc[f >> 1] = Tr<T>::mfma(vf[f % RING], pf[f & 1], oacc[f >> 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    dma_wait();
  }
  // ---- normalise and store: accumulator register r of block b <-> d = 32b + (r&3) + 8(r>>2) + 4hf of query l31
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q0 + l31 < N) {
    T* orow = o + (bt * N + (q0 + l31)) * FA_D + 4 * hf;
#pragma unroll
    for (int b = 0; b < FA_D / 32; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename Tr<T>::v4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (T)(oacc[b][g * 4 + j] * inv);
        *reinterpret_cast<typename Tr<T>::v4*>(orow + b * 32 + g * 8) = w;
      }
  }
}

}  // namespace cvvae

using namespace cvvae;

extern "C" int cvvae_attention_d512(int32_t dtype, const void* q, const void* k, const void* vt, void* o, int32_t batch, int32_t N,
                                    int64_t ldvt, float scale, void* stream) {
  if (!q || !k || !vt || !o || batch <= 0 || N <= 0 || ldvt < ((N + 31) / 32) * 32 || (ldvt % 8)) return CVVAE_EINVAL;
  if (dtype != CVVAE_F16 && dtype != CVVAE_BF16) return CVVAE_EUNSUPPORTED;
  const dim3 grid((unsigned)((N + 127) / 128), (unsigned)batch);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CVVAE_BF16)
    hipLaunchKernelGGL(flash_attn_d512_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)q, (const __bf16*)k, (const __bf16*)vt, (__bf16*)o, N,
                       (long long)ldvt, sl2);
  else
    hipLaunchKernelGGL(flash_attn_d512_kernel<_Float16>, grid, dim3(256), 0, s, (const _Float16*)q, (const _Float16*)k, (const _Float16*)vt,
                       (_Float16*)o, N, (long long)ldvt, sl2);
  return (int)hipGetLastError();
}
