// fp8 (OCP e4m3) "NT" GEMM for the ViT tower, BASELINE.json configs[3]:  C[M,N] = A8[M,K] . W8[N,K]^T  on the
// block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate) with all hardware block scales set to
// 1.0 (E8M0 0x7f) and the real scales applied in the epilogue:  out = acc * sa[m] * sw[n] + bias[n]
// (per-token activation scale, per-output-channel weight scale).  Operand / result lane maps were established on
// hardware by tools/probes/fp8probe2.hip: lane l holds row l&31 and 32 of its 64 k bytes; D as the 32x32 bf16 form.
// With unit scales any assignment of k bytes to (lane half, register half) is correct as long as both operands use the
// same one (this kernel gives lane half h the bytes [32h, 32h+32)).  The hardware's own order matters only for non-unit
// block scales (tools/probes/fp8probe3-5.hip): bytes 0-15 of lanes 0-31 are k 0..15, bytes 0-15 of lanes 32-63 are
// k 16..31, bytes 16-31 of lanes 0-31 are k 32..47, bytes 16-31 of lanes 32-63 are k 48..63; byte `opsel` of a lane's
// scale register scales block l>>5 (k 32*(l>>5) .. +31) of row l&31 of that operand.
//
// Same structure as gemm_persist.hip: persistent 256x256 tiles, 8 waves as 2(M) x 4(N), 4-slot LDS ring of 64-BYTE
// row stages (K = 64 fp8 elements per stage: the ring geometry, the LDS-DMA and the two-phase / half-phase-stagger
// schedule are byte-for-byte those of the bf16 kernel), cross-tile DMA prefetch, bf16 output through a wave-private
// LDS image.  Per stage and wave: 2 phases x 4 MFMAs of 64 cycles = the bf16 kernel's phase length at twice the K.
// The 16-B chunk swizzle gets one more term (subtile parity) so that the 32-row fragment reads are conflict-free.
#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int STG = 32768, WPART = 16384;
constexpr int RING = 4 * STG;               // 131072
constexpr int TR_OFF = RING;                // 8 x 4 KiB wave-private images
constexpr int LDS_BYTES = RING + 32768;     // 163840

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  else return u;
}

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at lds_off.
// Inline asm on purpose: behind the builtin, LLVM's waitcnt pass treats every later ds_read as possibly aliasing the
// pieces in flight and puts `s_waitcnt vmcnt(0)` in front of the fragment reads of EVERY phase -- a full drain of the
// prefetch ring twice per stage (it did so in this kernel; the ring's correctness is the hand-placed counted waits).
// (m0 is written; nothing else in this file uses it.)
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(lds_off));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

struct TileId { int m0, n0; };

__device__ __forceinline__ TileId decode_tile(int idx, int tiles_m, int tiles_n) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = idx & 7, pos = idx >> 3;
  const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;   // XCD-aware, bijective
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  return TileId{tm * BM, tn * BN};
}

// EPI: 0 = scale + bias (+ activation) -> bf16;  1 = scale + bias + residual (bf16, in place) -> bf16;
//      2 = scale + bias (+ activation), then * out_inv_scale[n] -> e4m3 (the next GEMM's operand, no bf16 round trip)
template <int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int r32 = lane & 31, h = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  const int G = gridDim.x;
  const size_t lda_b = (size_t)p.lda, ldw_b = (size_t)p.ldw;      // fp8: 1 byte per element
  const int kend = p.K;                      // bytes along K; one stage = 64 B; K % 256 == 0 (>= 4 stages)

  // LDS-DMA: instruction j of a part fills 1-KiB subtile 8j + w (16 rows x 64 B); LDS chunk lane&3 of row lane>>2
  // holds logical chunk (lane&3) ^ (2*(row>>3)) ^ (subtile & 1)
  const int lrow = 16 * w + (lane >> 2);
  const unsigned lchunk16 = (unsigned)(((lane & 3) ^ (((lane >> 5) & 1) << 1) ^ (w & 1)) * 16);
  const int dma_lds = w * 1024;
  const unsigned woff0 = (unsigned)(lrow * ldw_b) + lchunk16, woff1 = (unsigned)((128 + lrow) * ldw_b) + lchunk16;
  // fragment of a 32-row tile: lane (r32, h) reads the two 16-B chunks 2h, 2h+1 of row r32 (subtile r32>>4)
  const int rr = r32 & 15;
  const int swz = ((rr >> 3) << 1) ^ (r32 >> 4);
  const int rd0 = (r32 >> 4) * 1024 + rr * 64 + (((2 * h) ^ swz) << 4);
  const int rd1 = (r32 >> 4) * 1024 + rr * 64 + (((2 * h + 1) ^ swz) << 4);
  const int a_base = wr * 8 * 1024;                      // + slot*STG + mt*2048
  const int w_base = WPART + wc * 4 * 1024;              // + slot*STG + nt*2048

  // epilogue image: [32 rows][128 B] per wave, 16-B chunk index XOR row&7
  char* tr = smem + TR_OFF + w * 4096;
  const int tw_base = r32 * 128 + h * 8;
  const int tw_sw = r32 & 7;
  const int tr_base = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + k*1024: rows 8k + (lane>>3)
  const int row_l = lane >> 3;
#define TW_ADDR(c) (tr + tw_base + ((((c)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = decode_tile(idx, tiles_m, tiles_n);
  const char* Ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
  unsigned aoff0 = (unsigned)((min(cur.m0 + lrow, p.M - 1) - cur.m0) * lda_b) + lchunk16;
  unsigned aoff1 = (unsigned)((min(cur.m0 + 128 + lrow, p.M - 1) - cur.m0) * lda_b) + lchunk16;

#define STAGE_A(slot, blk, o0, o1, kbyte)                                                   \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (slot) * STG + dma_lds);                            \
    glds16((blk) + (kbyte), (o1), smem, (slot) * STG + 8192 + dma_lds);                     \
  } while (0)
#define STAGE_W(slot, blk, kbyte)                                                           \
  do {                                                                                      \
    glds16((blk) + (kbyte), woff0, smem, (slot) * STG + WPART + dma_lds);                   \
    glds16((blk) + (kbyte), woff1, smem, (slot) * STG + WPART + 8192 + dma_lds);            \
  } while (0)
#define LD_FRAG(dst, off)                                                                   \
  do {                                                                                      \
    const uint4 lo_ = *(const uint4*)(smem + (off) + rd0);                                  \
    const uint4 hi_ = *(const uint4*)(smem + (off) + rd1);                                  \
    dst = i32x8_t{(int)lo_.x, (int)lo_.y, (int)lo_.z, (int)lo_.w, (int)hi_.x, (int)hi_.y, (int)hi_.z, (int)hi_.w}; \
  } while (0)
#define LD_W(slot) _Pragma("unroll") for (int j = 0; j < 2; ++j) LD_FRAG(fb[j], (slot) * STG + w_base + j * 2048);
#define LD_A(slot, half) _Pragma("unroll") for (int i = 0; i < 2; ++i) LD_FRAG(fa[i], (slot) * STG + a_base + ((half) * 2 + i) * 2048);
#define MMA(half)                                                                           \
  do {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
      acc[(half) * 2 + i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j], fa[i], acc[(half) * 2 + i][j], 0, 0, \
                                                                              0, 0x7f7f7f7f, 0, 0x7f7f7f7f);        \
    __builtin_amdgcn_s_setprio(0);                                                          \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define STAGE(slot, ISSUE_W, ISSUE_A, WAIT_VM)                                              \
  do {                                                                                      \
    LD_W(slot) __builtin_amdgcn_sched_barrier(0); LD_A(slot, 0)                             \
    ISSUE_W;                                                                                \
    BARRIER(); WAIT_LDS(); MMA(0); BARRIER();                                               \
    LD_A(slot, 1)                                                                           \
    ISSUE_A;                                                                                \
    WAIT_VM;                                                                                \
    BARRIER(); WAIT_LDS(); MMA(1); BARRIER();                                               \
  } while (0)

  // Both parts of stage t+3 are issued in stage t; the counted wait leaves stages t+2 and t+3 in flight (4 x 2 pieces).
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
  // First two stages after an epilogue: what they need was issued before the epilogue's stores and vmcnt retires in
  // order, so the wait may leave the NST row stores of a complete tile outstanding as well (see gemm_persist.hip); one
  // opaque instruction for the compiler.
  constexpr int NST = EPI == 2 ? 8 : 16;     // row stores per wave and tile
#define VM_AFTER_EPILOGUE                                                                   \
  do {                                                                                      \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax > 0 ? 1 : 0);                     \
    relax = relax > 0 ? relax - 1 : 0;                                                      \
    if constexpr (NST == 16)                                                                \
      asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lf8vm8_%=\n\ts_waitcnt vmcnt(24)\n\ts_branch .Lf8end_%=\n"               \
                   ".Lf8vm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lf8end_%=:" : : "s"(sel_) : "memory", "scc");                            \
    else                                                                                    \
      asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lf8vm8_%=\n\ts_waitcnt vmcnt(16)\n\ts_branch .Lf8end_%=\n"               \
                   ".Lf8vm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lf8end_%=:" : : "s"(sel_) : "memory", "scc");                            \
  } while (0)

  // ---- cold prologue of the first tile ----
  STAGE_A(0, Ablk, aoff0, aoff1, 0); STAGE_W(0, Wblk, 0);
  STAGE_A(1, Ablk, aoff0, aoff1, 64); STAGE_W(1, Wblk, 64);
  STAGE_A(2, Ablk, aoff0, aoff1, 128); STAGE_W(2, Wblk, 128);
  VM8;
  int relax = 0;                             // stages of the coming tile that may leave the previous tile's stores in flight
  BARRIER();

  for (;;) {
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    i32x8_t fa[2], fb[2];

    if (wr == 1) BARRIER();                  // second wave row runs half a phase behind

    for (int kb = 0; kb < kend - 256; kb += 256) {
      STAGE(0, STAGE_W(3, Wblk, kb + 192), STAGE_A(3, Ablk, aoff0, aoff1, kb + 192), VM_AFTER_EPILOGUE);
      STAGE(1, STAGE_W(0, Wblk, kb + 256), STAGE_A(0, Ablk, aoff0, aoff1, kb + 256), VM_AFTER_EPILOGUE);
      STAGE(2, STAGE_W(1, Wblk, kb + 320), STAGE_A(1, Ablk, aoff0, aoff1, kb + 320), VM8);
      STAGE(3, STAGE_W(2, Wblk, kb + 384), STAGE_A(2, Ablk, aoff0, aoff1, kb + 384), VM8);
    }
    // ---- last four stages: the DMA crosses into the next tile (or re-fetches this one into dead slots) ----
    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    unsigned naoff0 = aoff0, naoff1 = aoff1;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
      naoff0 = (unsigned)((min(nxt.m0 + lrow, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
      naoff1 = (unsigned)((min(nxt.m0 + 128 + lrow, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
    }
    {
      const int kb = kend - 256;
      STAGE(0, STAGE_W(3, Wblk, kb + 192), STAGE_A(3, Ablk, aoff0, aoff1, kb + 192), VM_AFTER_EPILOGUE);
      STAGE(1, STAGE_W(0, Wnext, 0), STAGE_A(0, Anext, naoff0, naoff1, 0), VM_AFTER_EPILOGUE);
      STAGE(2, STAGE_W(1, Wnext, 64), STAGE_A(1, Anext, naoff0, naoff1, 64), VM8);
      STAGE(3, STAGE_W(2, Wnext, 128), STAGE_A(2, Anext, naoff0, naoff1, 128), VM8);
    }
    // pin the accumulators here: without a use in this block LLVM sinks the tail's 32 MFMAs below the conditional
    // barrier (all fragments live at once -> hundreds of spilled VGPRs)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (wr == 0) BARRIER();                  // re-align the two wave rows for the epilogue

    // ------------------------------- epilogue of tile `cur` -------------------------------
    // lane (r32, h) of m-tile mt holds row mw0 + mt*32 + r32, columns nb + nt*32 + 8g + 4h + (0..3) in acc[mt][nt][4g..4g+3]
    const int mw0 = cur.m0 + wr * 128;
    const int nb = cur.n0 + wc * 64;
    const size_t gcol = (size_t)nb + (lane & 7) * 8;
    // pass 1 (keeps registers low): acc <- acc * sw[n] + bias[n] / sa[m], so that pass 2 only multiplies by sa[m]
    float sa[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) sa[mt] = p.scale_a ? p.scale_a[min(mw0 + mt * 32 + r32, p.M - 1)] : 1.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int n = nb + (c >> 2) * 32 + (c & 3) * 8 + h * 4;
      const f32x4_t sw = *(const f32x4_t*)(p.scale_w + n);
      const f32x4_t bs = *(const f32x4_t*)(p.bias + n);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float rsa = __builtin_amdgcn_rcpf(sa[mt]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[mt][c >> 2][(c & 3) * 4 + e] = fmaf(acc[mt][c >> 2][(c & 3) * 4 + e], sw[e], bs[e] * rsa);
      }
    }
    uint4 rres[4];                           // residual rows: 4 x 16 B per 32-row block, one block in flight
#define LOAD_RES(k)                                                                           \
  do {                                                                                        \
    const int m_ = mw0 + (k) * 8 + row_l;                                                     \
    rres[(k) & 3] = uint4{0, 0, 0, 0};                                                        \
    if (m_ < p.M) rres[(k) & 3] = *(const uint4*)((const bf16_t*)p.resid + (size_t)m_ * p.ldo + gcol); \
  } while (0)
    if constexpr (EPI == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) LOAD_RES(k);
    }

    if constexpr (EPI == 2) {
      // fp8 image: [32 rows][80 B pitch] per wave (64 data bytes; the pitch keeps the dword writes 2-way conflicted at most),
      // the wave's 64 inverse output scales behind it
      float* isc = (float*)(tr + 2560);
      isc[lane] = p.out_inv_scale[nb + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int nt = c >> 2, g = c & 3;
          const int col = nt * 32 + g * 8 + h * 4;
          const f32x4_t is = *(const f32x4_t*)(isc + col);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = __builtin_amdgcn_fmed3f(act_apply_t<ACT>(acc[mt][nt][g * 4 + e] * sa[mt]) * is[e], -448.0f, 448.0f);
          int wd = 0;
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], wd, false);
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], wd, true);
          *(int*)(tr + r32 * 80 + col) = wd;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = k * 16 + (lane >> 2);
          const uint4 v = *(const uint4*)(tr + row * 80 + (lane & 3) * 16);
          const int m = mw0 + mt * 32 + row;
          if (m < p.M) *(uint4*)((char*)p.out + (size_t)m * p.ldo + nb + (lane & 3) * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (EPI == 1) {
        // residual rows of this 32-row block: row-major image -> fragment layout
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)(tr + k * 1024 + tr_base) = rres[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // cross-lane hand-off through the image
        if (mt + 1 < 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) LOAD_RES(mt * 4 + 4 + k);
        }
      }
      uint2 pk[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int nt = c >> 2, g = c & 3;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][g * 4 + e] * sa[mt];
        if constexpr (EPI == 1) {
          const uint2 rr = *(const uint2*)TW_ADDR(c);
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_t<ACT>(v[e]);
        }
        pk[c] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      if constexpr (EPI == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads done before the image is rewritten
#pragma unroll
      for (int c = 0; c < 8; ++c) *(uint2*)TW_ADDR(c) = pk[c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 v = *(const uint4*)(tr + k * 1024 + tr_base);
        const int m = mw0 + mt * 32 + k * 8 + row_l;
        if (m < p.M) *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // image reads done before the next block's writes
    }
    }
#undef LOAD_RES

    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    relax = (cur.m0 + BM <= p.M) ? 2 : 0;    // all 256 rows valid: every guarded row store above was issued
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext; aoff0 = naoff0; aoff1 = naoff1;
  }
}

template <int EPI, int ACT>
hipError_t launch_fp8(const GemmParams& p, hipStream_t stream) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_fp8_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    int dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, ACT>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

// A: fp8 [M][lda] (bytes), W: fp8 [N][ldw]; out bf16 [M][ldo]; scale_a [M], scale_w [N], bias [N] fp32.
// epi: EPI_STORE_BF16 (scale + bias + p.act), EPI_RESID (scale + bias + bf16 residual, may alias out) or
// EPI_STORE_FP8 (as STORE_BF16, then e4m3(value * out_inv_scale[n]) into out [M][ldo] BYTES).  scale_a NULL = 1.
hipError_t ce_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.M < 1 || p.N < BN || p.N % BN != 0 || p.K < 256 || p.K % 256 != 0) return hipErrorInvalidValue;
  if (p.lda % 16 != 0 || p.ldw % 16 != 0 || p.lda < p.K || p.ldw < p.K) return hipErrorInvalidValue;
  if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15) return hipErrorInvalidValue;
  if (!p.out || !p.scale_w || !p.bias) return hipErrorInvalidValue;
  if ((size_t)255 * p.lda + 64 >= 0x7fffffffull || (size_t)255 * p.ldw + 64 >= 0x7fffffffull) return hipErrorInvalidValue;
  if (epi == EPI_RESID) {
    if (!p.resid) return hipErrorInvalidValue;
    return launch_fp8<1, -1>(p, stream);
  }
  if (epi == EPI_STORE_FP8) {
    if (!p.out_inv_scale) return hipErrorInvalidValue;
    if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<2, CE_ACT_QUICK_GELU>(p, stream);
    if (p.act == CE_ACT_GELU_ERF) return launch_fp8<2, CE_ACT_GELU_ERF>(p, stream);
    return launch_fp8<2, -1>(p, stream);
  }
  if (epi != EPI_STORE_BF16) return hipErrorInvalidValue;
  if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<0, CE_ACT_QUICK_GELU>(p, stream);
  if (p.act == CE_ACT_GELU_ERF) return launch_fp8<0, CE_ACT_GELU_ERF>(p, stream);
  return launch_fp8<0, -1>(p, stream);
}
