// Training-side kernels of the YOLOv4 path on gfx950 (fp32): convolution weight gradient,
// zero-dilation for the data gradient of strided convolutions, and train-mode BatchNorm
// (+ activation, + residual) forward / backward.
//
// What they replace in the reference's training step:
//   cuDNN conv backward-filter / backward-data  (autograd of mmcv ConvModule, darknetcsp.py:15-35)
//   ATen batch_norm forward/backward in training mode + MishCudaFunction.backward (mish.py:27-36)
// The data gradient itself is the forward kernel again (conv_mfma_f32.hip) on dY with the
// weights transposed and flipped; for stride 2 dY is first zero-dilated (yv4_dilate2_fwd).
#include "yv4_common.h"

namespace yv4 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16_t(u32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ u32x4_t make_rsrc_t(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  u32x4_t v;
  v.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  v.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  v.z = __builtin_amdgcn_readfirstlane(bytes);
  v.w = 0x00020000u;
  return v;
}

// ---------------------------------------------------------------------------------
// Weight gradient:  dW[co][k] += sum_m dY[m][co] * A[m][k],  A = im2col(x), k = (kh,kw,ci).
// A workgroup owns a 64 (co) x 64 (k) tile of dW and one chunk of the M = N*Ho*Wo reduction;
// slices of 32 rows of dY and of A go global -> LDS by LDS-DMA (rows of 64 floats, read back
// with ds_read_b32 along the row, so no swizzle is needed), each wave accumulates a 32x32 tile
// on v_mfma_f32_32x32x2_f32 with the reduction index m as the MFMA K dimension, and the
// chunk's partial tile is added to dW with float atomics (dW must be zero on entry).
// ---------------------------------------------------------------------------------
// ---- element access for the three operand types (fp32, fp16, bf16): 4 consecutive channels ----
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct El;
template <> struct El<float> {
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct El<_Float16> {
  static __device__ __forceinline__ float4 ld4(const _Float16* p) {
    const f16x4_t v = *reinterpret_cast<const f16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  static __device__ __forceinline__ void st4(_Float16* p, float4 v) {
    f16x4_t o;
    o[0] = (_Float16)v.x; o[1] = (_Float16)v.y; o[2] = (_Float16)v.z; o[3] = (_Float16)v.w;
    *reinterpret_cast<f16x4_t*>(p) = o;
  }
};
template <> struct El<__bf16> {
  static __device__ __forceinline__ float4 ld4(const __bf16* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  static __device__ __forceinline__ void st4(__bf16* p, float4 v) {
    bf16x4_t o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = o;
  }
};

struct WgradArgs {
  const void* x;
  const void* dy;
  float* dw;
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int x_cs, x_co, dy_cs, dy_co;
  int M, K;
  int tiles_k, rows_per_chunk;
  int tiles = 0, chunks = 0, xcd_map = 0;   // see wgrad_tile_chunk
  int ablate = 0;                           // measurement build only (YV4_WFC_ABLATE)
  FastDiv fd_hw, fd_wo;     // m / (Ho*Wo), r / Wo: the per-slice row decode sits inside the pipelined loop
  // deterministic form: chunk c of the M reduction stores its partial dW to slab c of ws ([chunks][Cout][K], plain
  // stores); wgrad_reduce_kernel then adds the slabs to dw in chunk order.  ws == nullptr: float atomics into dw.
  float* ws = nullptr;
  long long ws_stride = 0;
};

constexpr int kWgRows = 32;   // reduction rows per slice

// Which (dW tile, reduction chunk) a workgroup serves.  All tiles of ONE chunk read the same rows of dY and of the
// activation (each its own columns, but whole 128-byte lines), and a layer whose dW has several tiles re-reads its
// operands once per tile column / row -- from HBM, when the tiles of a chunk sit on different XCDs: workgroups go to
// the 8 XCDs round-robin by linear id and every XCD has its own L2.  With xcd_map the grid is one-dimensional and
// workgroup L serves chunk 8 g + (L mod 8), tile j of it, with L / 8 = g * tiles + j: a chunk's tiles are neighbours in
// the launch order of ONE XCD, so the re-reads hit that XCD's L2, while the eight XCDs still sweep the reduction range
// side by side.  (Dealing each XCD one contiguous eighth of the (chunk, tile) pairs instead was measured too: the same
// gain on the 1x1 layers, but 20-30 % SLOWER on the HBM-bound few-channel layers at 304 / 608 pixels, whose XCDs then
// stream from eight distant regions.)  Batch 64, same box: 64->128 s2 @304 623 -> 461 us, 128->256 s2 @152 487 -> 379,
// 256->256 1x1 @38 41 -> 32; network 449 -> 478 TFLOP/s.  The chunk count is rounded to a multiple of 8 for it
// (wgrad_chunks).  Slabs and their summation order are indexed by the chunk, not by the workgroup: the result does
// not depend on the mapping.
__device__ __forceinline__ bool wgrad_tile_chunk(const int tiles, const int chunks, const int xcd_map, int& tile, int& chunk) {
  if (!xcd_map) {
    tile = (int)blockIdx.x;
    chunk = (int)blockIdx.y;
    return true;
  }
  const unsigned L = blockIdx.x;
  const unsigned j = L >> 3;
  const unsigned g = j / (unsigned)tiles;
  tile = (int)(j - g * (unsigned)tiles);
  chunk = (int)(g * 8u + (L & 7u));
  return chunk < chunks;
}
static inline dim3 wgrad_grid(long long tiles, long long chunks, int xcd_map) {
  if (!xcd_map) return dim3((unsigned)tiles, (unsigned)chunks);
  return dim3((unsigned)((chunks + 7) / 8 * 8 * tiles), 1u);
}

// T = float: rows of 64 floats (256 B), one LDS-DMA instruction of a wave covers 4 rows.
// T = _Float16 / __bf16: rows of 64 elements (128 B), one instruction covers 8 rows; the operands are
// widened to fp32 on the way from LDS to the MFMA (bf16 -> fp32 is a shift), so this form has the
// fp32 kernel's arithmetic and half its memory traffic.  (A v_mfma_f32_32x32x16 form needs both
// operands transposed on the way out of LDS -- ds_read_b64_tr_b16 -- and is next round's work.)
template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  constexpr int ES = (int)sizeof(T);
  constexpr int kChunkEl = 16 / ES;            // elements per 16-byte chunk: 4 or 8
  constexpr int kChunksPerRow = 64 / kChunkEl; // 16 or 8
  constexpr int kRowsPerDma = 64 / kChunksPerRow;   // rows covered by one wave-instruction: 4 or 8
  constexpr int kDmaPerWave = kWgRows / 4 / kRowsPerDma;  // instructions per wave per operand per slice: 2 or 1
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  // [2][32][64] dY slice, [2][32][64] A slice
  T* Ds = reinterpret_cast<T*>(smem_w);
  T* As = Ds + 2 * kWgRows * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const int tile_k = blockIdx.x % p.tiles_k;
  const int tile_c = blockIdx.x / p.tiles_k;
  const int co0 = tile_c * 64;
  const int k0 = tile_k * 64;
  const int m_lo = blockIdx.y * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;

  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_w;

  // staging: lane -> (row within the instruction, 16-byte chunk of the 64-element row)
  const int srow = lane / kChunksPerRow;
  const int chunk = lane % kChunksPerRow;
  const int dco = co0 + chunk * kChunkEl;
  const bool dco_ok = dco < p.Cout;          // Cout % kChunkEl == 0 is required by the host
  const int kk = k0 + chunk * kChunkEl;
  const bool k_ok = kk < p.K;
  const int tap = k_ok ? kk / p.Cin : 0;
  const int ci = kk - tap * p.Cin;
  const int kh = tap / p.KW;
  const int kw = tap - kh * p.KW;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  auto issue = [&](int m_base, int buf) {
#pragma unroll
    for (int q = 0; q < kDmaPerWave; ++q) {
      const int row0 = 8 * wave + kRowsPerDma * q;      // first row of this instruction within the slice
      const int m = m_base + row0 + srow;
      unsigned doff = kOOB, aoff = kOOB;
      if (m < m_hi) {
        if (dco_ok) doff = (unsigned)((((int64_t)m * p.dy_cs) + p.dy_co + dco) * ES);
        if (k_ok) {
          const int hw = p.Ho * p.Wo;
          const int n = fd_div(m, p.fd_hw);
          const int rm = m - n * hw;
          const int ho = fd_div(rm, p.fd_wo);
          const int wo = rm - ho * p.Wo;
          const int hi = ho * p.stride - p.pad + kh;
          const int wi = wo * p.stride - p.pad + kw;
          if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
            aoff = (unsigned)(((((int64_t)n * p.H + hi) * p.W + wi) * p.x_cs + p.x_co + ci) * ES);
        }
      }
      const unsigned lrow = (unsigned)((buf * kWgRows + row0) * 64 * ES);
      lds_dma16_t(rsD, lds_base + lrow, doff, 0u);
      lds_dma16_t(rsX, lds_base + (unsigned)(2 * kWgRows * 64 * ES) + lrow, aoff, 0u);
    }
  };

  const int nslices = (m_hi - m_lo + kWgRows - 1) / kWgRows;
  issue(m_lo, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < nslices; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslices) issue(m_lo + (s + 1) * kWgRows, buf ^ 1);
    const T* ds = Ds + buf * kWgRows * 64 + wm * 32 + r;   // dY^T operand: [m][co]
    const T* as = As + buf * kWgRows * 64 + wn * 32 + r;   // A operand:    [m][k]
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < kWgRows / 2; ++t) {
      const float a = (float)ds[(2 * t + h) * 64];
      const float b = (float)as[(2 * t + h) * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // D[row = co][col = k]: row = (e&3) + 8*(e>>2) + 4*h, col = r
  const int kcol = k0 + wn * 32 + r;
  if (kcol < p.K) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (co < p.Cout) {
        if (p.ws) p.ws[(size_t)blockIdx.y * p.ws_stride + (size_t)co * p.K + kcol] = acc[e];
        else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[e]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Weight gradient on the 16-bit MFMA (v_mfma_f32_32x32x16_{f16,bf16}).
// Both operands of  dW[co][k] = sum_m dY[m][co] * A[m][k]  have the reduction index m as their
// SLOW memory dimension (NHWC: channels contiguous), while an MFMA lane needs 8 consecutive m of
// one column.  The slices therefore go global -> LDS row-major by LDS-DMA ([m][128 columns], 256-byte
// rows) and come out transposed through ds_read_b64_tr_b16: per 16-lane group a 4 (m) x 16 (column)
// block, lane i receiving column i -- two such reads are one lane's 8-element MFMA operand.
// Chunk swizzle (cdna_hip_programming.md T10, image (b)): the 16-byte chunk ch of row `row` lives at
// ch ^ (((row&3)<<2) | ((row>>2)&3)), applied on the DMA source side and in the read addresses;
// without it the four rows of a block share 16 banks.
// A workgroup (4 waves, 2x2, each 64 co x 64 k = 4 accumulator tiles) owns a 128 x 128 tile of dW
// and one chunk of the M reduction, 64 rows per slice, double-buffered (64 KB of LDS).
// ---------------------------------------------------------------------------------
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_w __attribute__((ext_vector_type(8)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int kWhRows = 64;    // reduction rows per slice
constexpr int kWhTile = 128;   // dW tile edge (co and k)

template <bool BF16, int NBUF>
__global__ __launch_bounds__(256, NBUF == 2 ? 2 : 1) void conv_wgrad_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  extern __shared__ __attribute__((aligned(16))) char smem_wh[];
  constexpr int kRowB = 256;                                  // 128 columns x 2 bytes
  constexpr int kOpBytes = kWhRows * kRowB;                   // one operand, one buffer: 16 KB
  // layout: [buf][operand (0 = dY, 1 = A)][64 rows][256 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int tile_id, chunk;
  if (!wgrad_tile_chunk(p.tiles, p.chunks, p.xcd_map, tile_id, chunk)) return;
  const int tile_k = tile_id % p.tiles_k;
  const int tile_c = tile_id / p.tiles_k;
  const int co0 = tile_c * kWhTile;
  const int k0 = tile_k * kWhTile;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;

  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_wh;

  // ---- staging: instruction q of this wave fills rows 16*wave + 4q + lane/16, physical chunk lane%16
  const int srow = lane >> 4;
  const int pc = lane & 15;
  int d_col[4];            // dY column offset (elements) of the logical chunk, or -1
  int a_tap[4], a_kh[4], a_kw[4], a_ci[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lc = pc ^ ((srow << 2) | q);                    // swizzle: row&3 = srow, (row>>2)&3 = q
    const int co = co0 + lc * 8;
    d_col[q] = co < p.Cout ? co : -1;                          // Cout % 8 == 0 (host)
    const int kk = k0 + lc * 8;
    if (kk < p.K) {
      const int tap = kk / p.Cin;
      a_tap[q] = tap;
      a_ci[q] = kk - tap * p.Cin;
      a_kh[q] = tap / p.KW;
      a_kw[q] = tap - a_kh[q] * p.KW;
    } else {
      a_tap[q] = -1; a_ci[q] = 0; a_kh[q] = 0; a_kw[q] = 0;
    }
  }

  // (a slice past the end of the chunk is issued all the same, every lane out of range: the count of outstanding
  // instructions the waits below rely on stays fixed)
  // No branches in here: every lane decodes its four rows the same way and SELECTS between its offset and the
  // out-of-range one (the nested ifs and the wrap loop of the first form compiled to a dozen divergent branches per
  // slice in front of the fragment reads).
  auto issue = [&](int m_base, int buf) {
    const int hw = p.Ho * p.Wo;
    const unsigned lrow0 = (unsigned)(buf * 2 * kOpBytes + (16 * wave) * kRowB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m_base + 16 * wave + srow + 4 * q;
      const int n = fd_div(m, p.fd_hw);
      const int rm = m - n * hw;
      const int ho = fd_div(rm, p.fd_wo);
      const int wo = rm - ho * p.Wo;
      const bool rowok = m < m_hi;
      const int hi = ho * p.stride - p.pad + a_kh[q];
      const int wi = wo * p.stride - p.pad + a_kw[q];
      const bool dok = rowok && d_col[q] >= 0;
      const bool aok = rowok && a_tap[q] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const unsigned dval = (unsigned)m * (unsigned)(p.dy_cs * 2) + (unsigned)((p.dy_co + d_col[q]) * 2);
      const unsigned aval = (unsigned)((n * p.H + hi) * p.W + wi) * (unsigned)(p.x_cs * 2) + (unsigned)((p.x_co + a_ci[q]) * 2);
      const unsigned doff = dok ? dval : kOOB;
      const unsigned aoff = aok ? aval : kOOB;
      const unsigned lrow = lrow0 + (unsigned)(4 * q * kRowB);
      lds_dma16_t(rsD, lds_base + lrow, doff, 0u);
      lds_dma16_t(rsX, lds_base + (unsigned)kOpBytes + lrow, aoff, 0u);
    }
  };

  // ---- transposed fragment reads ----
  // lane = 16g + i: h = g>>1 (k half of the MFMA step), colhalf = g&1; inside the group lane 4q'+pp
  // addresses row q' of the block, columns 4pp..4pp+3
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  // byte offset inside an operand buffer of (block row m0 + qq, column c0 + 4pp), m0 = 16s + 8hh + 4j:
  //   256*(m0+qq) + 16*((c0/8 + (pp>>1)) ^ ((qq<<2) | ((2hh + j)&3))) + 8*(pp&1)
  auto frag_addr = [&](int col_base, int s, int j) -> unsigned {
    const int m0 = 16 * s + 8 * hh + 4 * j;
    const int chunk = (col_base + 16 * colhalf) / 8 + (pp >> 1);
    const int swz = (qq << 2) | ((2 * hh + j) & 3);
    return (unsigned)(kRowB * (m0 + qq) + 16 * (chunk ^ swz) + 8 * (pp & 1));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // NBUF slice buffers, NBUF - 1 slices of LDS-DMA in flight.  A 128 x 128 tile spends ~500 cycles of MFMA on a slice
  // whose 32 KB take the memory system several times that to deliver: the kernel runs at (bytes in flight) / latency.
  // NBUF = 2 (two workgroups per CU, each waiting out its one outstanding slice) keeps 2 x 32 KB in flight per CU,
  // NBUF = 4 (one workgroup, 128 KB of LDS) three slices -- and never drains the queue: the wait in front of slice s
  // leaves the (NBUF - 2) younger slices outstanding (8 DMA instructions per wave and slice).
  const int nslices = (m_hi - m_lo + kWhRows - 1) / kWhRows;
#pragma unroll
  for (int s0 = 0; s0 < NBUF - 1; ++s0) issue(m_lo + s0 * kWhRows, s0);
  for (int sl = 0; sl < nslices; ++sl) {
    const int buf = sl % NBUF;
    if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NBUF == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (NBUF == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // slice sl is in LDS; every wave is done with slice sl - 1
    asm volatile("" ::: "memory");
    issue(m_lo + (sl + NBUF - 1) * kWhRows, (sl + NBUF - 1) % NBUF);        // into the buffer slice sl - 1 left
    char* dbuf = smem_wh + buf * 2 * kOpBytes;
    char* abuf = dbuf + kOpBytes;
    __builtin_amdgcn_s_setprio(1);
    // two fragment sets: the eight reads of step s + 1 are issued in front of the four MFMAs of step s (one set, as the
    // compiler schedules the plain loop, makes every step wait out a fresh LDS round trip)
    s16x8_t fa[2][2], fb[2][2];
#define YV4_WH_LOAD(SET, S)                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                   \
      const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(dbuf + frag_addr(wm * 64 + t * 32, S, 0))); \
      const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(dbuf + frag_addr(wm * 64 + t * 32, S, 1))); \
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(abuf + frag_addr(wn * 64 + t * 32, S, 0))); \
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(abuf + frag_addr(wn * 64 + t * 32, S, 1))); \
      fa[SET][t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);                                          \
      fb[SET][t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);                                          \
    }
#define YV4_WH_MFMA(SET)                                                                                              \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                     \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                 \
        if (BF16)                                                                                                     \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET][a]),               \
                                                              __builtin_bit_cast(bf16x8_w, fb[SET][b]), acc[a][b], 0, 0, 0); \
        else                                                                                                          \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET][a]),                 \
                                                             __builtin_bit_cast(f16x8_w, fb[SET][b]), acc[a][b], 0, 0, 0);   \
      }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
    static_assert(kWhRows / 16 == 4, "the step schedule below is written for four 16-row steps");
    YV4_WH_LOAD(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WH_LOAD(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WH_MFMA(0);
    YV4_WH_LOAD(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WH_MFMA(1);
    YV4_WH_LOAD(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WH_MFMA(0);
    YV4_WH_MFMA(1);
#undef YV4_WH_MFMA
#undef YV4_WH_LOAD
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range slices issued past the end
  // D[row = co][col = k]: row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31
  const int r = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kcol = k0 + wn * 64 + b * 32 + r;
      if (kcol >= p.K) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
        if (co < p.Cout) {
          if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[a][b][e];
          else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[a][b][e]);
        }
      }
    }
}

// ---------------------------------------------------------------------------------
// The kernel above with its staging arithmetic removed from the lanes (round 5).  Per 64-row slice and wave the loop
// above issues 16 MFMAs (512 matrix-pipe cycles) and 119 VALU instructions of which 40 are 32-bit integer multiplies
// (quarter rate: 16 issue cycles each) -- the row decode (two divisions by invariant divisors) and the byte offsets of
// four rows per lane, rebuilt from the row index every slice: ~960 cycles of vector issue per 512 of matrix work, two
// waves per SIMD.  Here:
//   * LINEAR (1x1, stride 1, no padding -- 27 of YOLOv4-L's layers): both operands' offsets advance by a constant per
//     slice and are range-checked as offsets against lane-constant limits: an add, a compare and a select per piece;
//   * otherwise one wave decodes each of the slice's 64 rows ONCE (64 lanes = 64 rows) two slices ahead and leaves
//     {byte offset of the row's window origin, (hi0, wi0)} in an LDS table beside the slice buffers; a lane reads the
//     entries of its four rows, adds its taps' lane-constant offset and checks the window bounds -- no division, no multiply.
// Same tiles, same MFMAs in the same order, same slab output: dW is bit-identical to the kernel above.
// ---------------------------------------------------------------------------------
constexpr int kWhLds2 = 2 * 2 * kWhRows * 256;       // two slice buffers
constexpr int kWhTabBytes = kWhRows * 8;             // one row table
constexpr int kWhLdsV2 = kWhLds2 + 3 * kWhTabBytes;       // three tables: a slice's entries are read a slice ahead of its DMA

template <bool BF16, bool LINEAR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_v2_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) u32x2_t* lds_u2_t;
  extern __shared__ __attribute__((aligned(16))) char smem_wv[];
  constexpr int kRowB = 256;
  constexpr int kOpBytes = kWhRows * kRowB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int tile_id, chunk;
  if (!wgrad_tile_chunk(p.tiles, p.chunks, p.xcd_map, tile_id, chunk)) return;
  const int tile_k = tile_id % p.tiles_k;
  const int tile_c = tile_id / p.tiles_k;
  const int co0 = tile_c * kWhTile;
  const int k0 = tile_k * kWhTile;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;

  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_wv;

  // ---- staging: instruction q of this wave fills rows 16*wave + 4q + lane/16, physical chunk lane%16
  const int srow = lane >> 4;
  const int pc = lane & 15;
  const unsigned d_step = (unsigned)(kWhRows * p.dy_cs * 2), x_step = (unsigned)(kWhRows * p.x_cs * 2);
  unsigned d_off[4], d_lim[4];       // dY: byte offset of this lane's 16 bytes at slice 0, and its limit (0: never)
  unsigned a_off[4], a_lim[4];       // LINEAR: the same for the activation
  unsigned a_tap[4];                 // general: byte offset of the lane's (tap, channel chunk) from the row's window origin
  int a_kh[4], a_kw[4];              // general: the tap (kh = -30000 for a column beyond K: never inside the map)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lc = pc ^ ((srow << 2) | q);                    // swizzle: row&3 = srow, (row>>2)&3 = q
    const int row = 16 * wave + srow + 4 * q;
    const int co = co0 + lc * 8;
    const unsigned cb = (unsigned)((p.dy_co + co) * 2);
    d_off[q] = (unsigned)(m_lo + row) * (unsigned)(p.dy_cs * 2) + cb;
    d_lim[q] = co < p.Cout ? (unsigned)m_hi * (unsigned)(p.dy_cs * 2) + cb : 0u;
    const int kk = k0 + lc * 8;
    int tap = 0, ci = 0;
    const bool kok = kk < p.K;
    if (kok) { tap = kk / p.Cin; ci = kk - tap * p.Cin; }
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const unsigned xb_ = (unsigned)((p.x_co + ci) * 2);
    if (LINEAR) {      // tap 0 only; input pixel = output pixel
      a_off[q] = (unsigned)(m_lo + row) * (unsigned)(p.x_cs * 2) + xb_;
      a_lim[q] = kok ? (unsigned)m_hi * (unsigned)(p.x_cs * 2) + xb_ : 0u;
    } else {
      a_kh[q] = kok ? kh : -30000;
      a_kw[q] = kw;
      a_tap[q] = (unsigned)(kh * p.W + kw) * (unsigned)(p.x_cs * 2) + xb_;
    }
  }
  // general form: the row table.  Entry of slice row r: {byte offset of input pixel (n, ho*s - p, wo*s - p) -- may be in
  // front of the map: arithmetic modulo 2^32 --, (hi0 << 16) | (wi0 & 0xFFFF)}; rows past the chunk get hi0 = -30000.
  int t_m = m_lo + lane;             // (wave 0) the row this lane decodes next
  auto table = [&](int sl) {
    if (LINEAR || wave != 0) return;
    unsigned off0 = 0u;
    int hi0 = -30000, wi0 = 0;
    if (t_m < m_hi) {
      const int n = fd_div(t_m, p.fd_hw);
      const int rm = t_m - n * (p.Ho * p.Wo);
      const int ho = fd_div(rm, p.fd_wo);
      const int wo = rm - ho * p.Wo;
      hi0 = ho * p.stride - p.pad;
      wi0 = wo * p.stride - p.pad;
      off0 = (unsigned)((n * p.H + hi0) * p.W + wi0) * (unsigned)(p.x_cs * 2);
    }
    u32x2_t ent;
    ent.x = off0;
    ent.y = ((unsigned)hi0 << 16) | ((unsigned)wi0 & 0xFFFFu);
    *(lds_u2_t)(smem_wv + kWhLds2 + (sl % 3) * kWhTabBytes + lane * 8) = ent;
    t_m += kWhRows;
  };
  // the table entries of this lane's four rows of slice `sl`: fetched a whole slice ahead of the DMA that uses them (an LDS
  // round trip in front of every slice's issue cost the streaming layers 7 %)
  u32x2_t te[4] = {};
  auto fetch = [&](int sl) {
    if (LINEAR) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      te[q] = *(lds_u2_t)(smem_wv + kWhLds2 + (sl % 3) * kWhTabBytes + (16 * wave + srow + 4 * q) * 8);
  };
  auto issue = [&](int sl) {
    const int buf = sl & 1;
    const unsigned lrow0 = (unsigned)(buf * 2 * kOpBytes + (16 * wave) * kRowB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned lrow = lrow0 + (unsigned)(4 * q * kRowB);
      lds_dma16_t(rsD, lds_base + lrow, d_off[q] < d_lim[q] ? d_off[q] : kOOB, 0u);
      d_off[q] += d_step;
      unsigned aoff;
      if (LINEAR) {
        aoff = a_off[q] < a_lim[q] ? a_off[q] : kOOB;
        a_off[q] += x_step;
      } else {
        const int hi = ((int)te[q].y >> 16) + a_kh[q];
        const int wi = (int)(short)(te[q].y & 0xFFFFu) + a_kw[q];
        aoff = ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) ? te[q].x + a_tap[q] : kOOB;
      }
      lds_dma16_t(rsX, lds_base + (unsigned)kOpBytes + lrow, aoff, 0u);
    }
  };

  // ---- transposed fragment reads (conv_wgrad_h16_kernel)
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  auto frag_addr = [&](int col_base, int s_, int j) -> unsigned {
    const int m0 = 16 * s_ + 8 * hh + 4 * j;
    const int chunk_ = (col_base + 16 * colhalf) / 8 + (pp >> 1);
    const int swz = (qq << 2) | ((2 * hh + j) & 3);
    return (unsigned)(kRowB * (m0 + qq) + 16 * (chunk_ ^ swz) + 8 * (pp & 1));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nslices = (m_hi - m_lo + kWhRows - 1) / kWhRows;
  table(0);
  table(1);
  table(2);
  if (!LINEAR) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  fetch(0);
  issue(0);
  fetch(1);
  for (int sl = 0; sl < nslices; ++sl) {
    const int buf = sl & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // slice sl is in LDS (and table sl + 2); every wave is done with slice sl - 1
    asm volatile("" ::: "memory");
    issue(sl + 1);                                   // into the buffer slice sl - 1 left; its table entries are in registers
    table(sl + 3);                                   // into the table whose entries (slice sl) every wave fetched two barriers ago
    fetch(sl + 2);                                   // written during slice sl - 1, visible since the barrier above
    char* dbuf = smem_wv + buf * 2 * kOpBytes;
    char* abuf = dbuf + kOpBytes;
    __builtin_amdgcn_s_setprio(1);
    s16x8_t fa[2][2], fb[2][2];
#define YV4_WV_LOAD(SET, S)                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                   \
      const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(dbuf + frag_addr(wm * 64 + t * 32, S, 0))); \
      const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(dbuf + frag_addr(wm * 64 + t * 32, S, 1))); \
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(abuf + frag_addr(wn * 64 + t * 32, S, 0))); \
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(abuf + frag_addr(wn * 64 + t * 32, S, 1))); \
      fa[SET][t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);                                          \
      fb[SET][t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);                                          \
    }
#define YV4_WV_MFMA(SET)                                                                                              \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                     \
      _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                 \
        if (BF16)                                                                                                     \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET][a]),               \
                                                              __builtin_bit_cast(bf16x8_w, fb[SET][b]), acc[a][b], 0, 0, 0); \
        else                                                                                                          \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET][a]),                 \
                                                             __builtin_bit_cast(f16x8_w, fb[SET][b]), acc[a][b], 0, 0, 0);   \
      }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
    YV4_WV_LOAD(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WV_LOAD(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WV_MFMA(0);
    YV4_WV_LOAD(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WV_MFMA(1);
    YV4_WV_LOAD(1, 3);
    __builtin_amdgcn_sched_barrier(0);
    YV4_WV_MFMA(0);
    YV4_WV_MFMA(1);
#undef YV4_WV_MFMA
#undef YV4_WV_LOAD
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range slice issued past the end
  const int r = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int kcol = k0 + wn * 64 + b * 32 + r;
      if (kcol >= p.K) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
        if (co < p.Cout) {
          if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[a][b][e];
          else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[a][b][e]);
        }
      }
    }
}

// ---------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride-1 / pad-1 layers with Cin % 128 == 0 (71 % of YOLOv4-L's weight-gradient FLOPs):
// the three kw taps of one (kh, 128-channel chunk) share ONE LDS image of the slice's source pixels.
//   dW[co][kh][kw][ci] = sum_m dY[m][co] * X[m + (kh - 1) W + (kw - 1)][ci]      (flattened pixel index m; borders masked)
// The kernel above fetches 32 KB per 64-row slice for a 128 x 128 tile of dW (64 FLOP per byte of LDS fill, the regime
// in which the forward tiles sit at the L2 -> LDS limit).  Here an 8-wave workgroup owns 128 co x (3 kw x 128 ci) of dW
// and one chunk of the M reduction: per slice the 64 rows of dY and the 66 source pixels of X (one image for all three
// kw: operand row = reduction row + kw) are 32.5 KB of fill for 6.3 MFLOP -- 190 FLOP per byte -- and a wave (64 co x 32
// ci x 3 kw = six accumulators) needs ten transposed reads per six MFMAs instead of eight per four.  What a shifted row
// must not see (left / right image border, rows above / below, the neighbouring image) is masked per LANE: a lane of a
// ds_read_b64_tr_b16 supplies the address of ONE reduction row, so redirecting it to a zero row zeroes that row's
// contribution for every column of the transposed block.  Four slice buffers, three slices of LDS-DMA in flight, one
// barrier per slice placed in front of the LAST 16-row step so that the next slice's first fragments are read while
// that step's MFMAs run.  Same chunked, deterministic output as above (slab per chunk + wgrad_reduce_kernel).
// ---------------------------------------------------------------------------------
constexpr int kW3Threads = 512;
constexpr int kW3Rows = 64;                        // reduction rows per slice
constexpr int kW3XRows = 68;                       // 66 source pixels + one DMA group of 4; rows 66, 67 are only ever zero
constexpr int kW3ZeroRow = 66;
constexpr int kW3BufBytes = (kW3Rows + kW3XRows) * 256;
constexpr int kW3NBuf = 4;
constexpr int kW3Lds = kW3NBuf * kW3BufBytes;      // 135 168 B: one workgroup per CU

#ifdef YV4_MEASURE   // the FIRST form of the 3x3 weight gradient: the measurement build's A/B partner of the second form (same bits); the product takes the generic 16-bit kernel where the second form does not apply
template <bool BF16>
__global__ __launch_bounds__(kW3Threads, 2) void conv_wgrad3x3_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  extern __shared__ __attribute__((aligned(16))) char smem_w3[];
  constexpr int kRowB = 256;
  constexpr int kDBytes = kW3Rows * kRowB;           // dY part of a buffer; the X image follows it
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 2;                          // co half (64 rows of dW)
  const int wk = wave & 3;                           // ci slab (32 columns per kw)

  // tile: (co tile, kh, ci tile), ci fastest
  const int tiles_ci = p.Cin >> 7;
  int tile, chunk;
  if (!wgrad_tile_chunk(p.tiles, p.chunks, p.xcd_map, tile, chunk)) return;
  const int tci = tile % tiles_ci;
  const int kh = (tile / tiles_ci) % 3;
  const int tco = tile / (3 * tiles_ci);
  const int co0 = tco * 128, ci0 = tci * 128;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;
  const int NHW = p.N * p.H * p.W;

  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_w3;

  // ---- staging: a DMA instruction covers 4 rows x 16 chunks; wave w fills rows 8w .. 8w+7 of dY and of the X image,
  // wave 0 also the 17th group of the image (rows 64 .. 67: pixels 64, 65 + two zero rows)
  const int srow = lane >> 4;
  const int pc = lane & 15;
  auto swz_of = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  int d_col[2], x_col[3], x_row[3];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * wave + 4 * q + srow;
    const int lc = pc ^ swz_of(row);
    const int co = co0 + lc * 8;
    d_col[q] = co < p.Cout ? co : -1;
    x_col[q] = ci0 + lc * 8;
    x_row[q] = row;
  }
  {
    const int row = 64 + srow;
    x_col[2] = ci0 + (pc ^ swz_of(row)) * 8;
    x_row[2] = row;
  }
  const int x_shift = (kh - 1) * p.W - 1;            // image row ir <-> pixel m_slice + ir + x_shift
  auto issue = [&](int sl, int nsl) {
    const int buf = sl & (kW3NBuf - 1);
    const int m_base = m_lo + sl * kW3Rows;
    const bool live = sl < nsl;
    const unsigned lb = lds_base + (unsigned)(buf * kW3BufBytes + 8 * wave * kRowB);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int m = m_base + 8 * wave + 4 * q + srow;
      unsigned doff = kOOB;
      if (live && m < m_hi && d_col[q] >= 0) doff = (unsigned)((((int64_t)m * p.dy_cs) + p.dy_co + d_col[q]) * 2);
      lds_dma16_t(rsD, lb + (unsigned)(4 * q * kRowB), doff, 0u);
      const int pix = m_base + x_row[q] + x_shift;
      unsigned xoff = kOOB;
      if (live && (unsigned)pix < (unsigned)NHW) xoff = (unsigned)((((int64_t)pix * p.x_cs) + p.x_co + x_col[q]) * 2);
      lds_dma16_t(rsX, lb + (unsigned)(kDBytes + 4 * q * kRowB), xoff, 0u);
    }
    if (wave == 0) {
      const int pix = m_base + x_row[2] + x_shift;
      unsigned xoff = kOOB;
      if (live && x_row[2] < 66 && (unsigned)pix < (unsigned)NHW) xoff = (unsigned)((((int64_t)pix * p.x_cs) + p.x_co + x_col[2]) * 2);
      lds_dma16_t(rsX, lds_base + (unsigned)(buf * kW3BufBytes + kDBytes + 64 * kRowB), xoff, 0u);
    }
  };

  // ---- transposed fragment reads (see conv_wgrad_h16_kernel): lane = 16 g + 4 qq + pp supplies row (block + qq),
  // columns 4 pp .. 4 pp + 3 of its 16-column half
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  auto row_addr = [&](int row, int col_base) -> unsigned {
    const int chunk = (col_base + 16 * colhalf) / 8 + (pp >> 1);
    return (unsigned)(kRowB * row + 16 * (chunk ^ swz_of(row)) + 8 * (pp & 1));
  };
  unsigned d_rd[2][4][2];                            // dY: [co tile a][step s][j]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) d_rd[a][s][j] = row_addr(16 * s + 8 * hh + 4 * j + qq, wc * 64 + a * 32);
  const unsigned zero_rd = (unsigned)(kDBytes + kW3ZeroRow * kRowB);

  f32x16 acc[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nsl = (m_hi - m_lo + kW3Rows - 1) / kW3Rows;
  // border masks of this lane's eight reduction rows of a slice: bit (s * 2 + j) * 3 + kw set = row contributes to tap kw
  auto slice_masks = [&](int sl) -> unsigned {
    unsigned mk = 0u;
    const int hw = p.H * p.W;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m_lo + sl * kW3Rows + 16 * s + 8 * hh + 4 * j + qq;
        unsigned b3 = 0u;
        if (m < m_hi) {
          const int n = fd_div(m, p.fd_hw);
          const int rm = m - n * hw;
          const int ho = fd_div(rm, p.fd_wo);
          const int wo = rm - ho * p.W;
          if ((unsigned)(ho + kh - 1) < (unsigned)p.H)
            b3 = (wo > 0 ? 1u : 0u) | 2u | (wo + 1 < p.W ? 4u : 0u);
        }
        mk |= b3 << ((s * 2 + j) * 3);
      }
    return mk;
  };

  // measurement-only bits (YV4_W3_ABLATE): 1 no MFMAs, 2 no fragment reads, 4 no DMA after the prologue, 8 no border
  // masks, 16 no output
  s16x8_t fa[2][2] = {}, fb[2][3] = {};              // fragment sets: step s computes from set s & 1
#define YV4_W3_LOAD(SET, BUFP, S, MK)                                                                         \
  if (!YV4_ABLATE(p.ablate, 2)) {                                                                             \
    const char* db_ = (BUFP);                                                                                 \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                           \
      const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(db_ + d_rd[a][S][0]));           \
      const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(db_ + d_rd[a][S][1]));           \
      fa[SET][a] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);                                   \
    }                                                                                                         \
    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                                        \
      const unsigned r0_ = (((MK) >> (((S) * 2 + 0) * 3 + kw)) & 1u)                                          \
          ? (unsigned)kDBytes + row_addr(16 * (S) + 8 * hh + qq + kw, wk * 32) : zero_rd;                     \
      const unsigned r1_ = (((MK) >> (((S) * 2 + 1) * 3 + kw)) & 1u)                                          \
          ? (unsigned)kDBytes + row_addr(16 * (S) + 8 * hh + 4 + qq + kw, wk * 32) : zero_rd;                 \
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(db_ + r0_));                      \
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(db_ + r1_));                      \
      fb[SET][kw] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);                                  \
    }                                                                                                         \
  }
#define YV4_W3_MFMA(SET)                                                                                      \
  {                                                                                                           \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                             \
      _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                                      \
        if (YV4_ABLATE(p.ablate, 1)) { acc[a][kw][0] += __builtin_bit_cast(float, (int)(fa[SET][a][0] + fb[SET][kw][0])); continue; } \
        if (BF16)                                                                                             \
          acc[a][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET][a]),      \
                                                               __builtin_bit_cast(bf16x8_w, fb[SET][kw]), acc[a][kw], 0, 0, 0); \
        else                                                                                                  \
          acc[a][kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET][a]),        \
                                                              __builtin_bit_cast(f16x8_w, fb[SET][kw]), acc[a][kw], 0, 0, 0); \
      }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
  // pieces per slice: 4 (5 on wave 0).  In slice t the wave issues DMA(t + 3) BEFORE the wait in front of the last
  // step, where it needs its own DMA(t + 1) landed: DMA(t + 2) and DMA(t + 3) may stay in flight.
#define YV4_W3_WAIT()                                                                                         \
  {                                                                                                           \
    if (wave == 0) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");                               \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                          \
  }

  issue(0, nsl);
  issue(1, nsl);
  issue(2, nsl);
  YV4_W3_WAIT();                                      // DMA(0) landed (newer: 1, 2)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned mk = slice_masks(0);
  YV4_W3_LOAD(0, smem_w3, 0, mk);
  for (int sl = 0; sl < nsl; ++sl) {
    const char* bufp = smem_w3 + (sl & (kW3NBuf - 1)) * kW3BufBytes;
    const char* nbufp = smem_w3 + ((sl + 1) & (kW3NBuf - 1)) * kW3BufBytes;
    const unsigned mkn = YV4_ABLATE(p.ablate, 8) ? 0xFFFFFFu : slice_masks(sl + 1);
    YV4_W3_LOAD(1, bufp, 1, mk);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3_MFMA(0);
    if (!YV4_ABLATE(p.ablate, 4)) issue(sl + 3, nsl);   // into the buffer slice sl - 1 read (freed by the previous barrier)
    else issue(nsl, nsl);                             // (the counted waits need the instruction count: all lanes out of range)
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3_LOAD(0, bufp, 2, mk);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3_MFMA(1);
    YV4_W3_LOAD(1, bufp, 3, mk);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3_MFMA(0);
    YV4_W3_WAIT();                                    // own DMA(sl + 1) landed; every read of slice sl has returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    YV4_W3_LOAD(0, nbufp, 0, mkn);                    // (beyond the last slice: zero-filled buffers, never used)
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3_MFMA(1);
    mk = mkn;
  }
#undef YV4_W3_WAIT
#undef YV4_W3_MFMA
#undef YV4_W3_LOAD
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the dummy tail DMAs must land before the LDS is released

  // D[row = co][col = ci]: row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31
  const int r = lane & 31, h5 = lane >> 5;
  if (YV4_ABLATE(p.ablate, 16) && acc[0][0][0] != 123.f) return;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int kcol = (kh * 3 + kw) * p.Cin + ci0 + wk * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wc * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
        if (co < p.Cout) {
          if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[a][kw][e];
          else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[a][kw][e]);
        }
      }
    }
}
#endif  // YV4_MEASURE

// ---------------------------------------------------------------------------------
// The same kernel with its loop overhead removed (round 5).  The disassembly of the kernel above has, per 64-row slice and
// wave, 24 MFMAs (768 matrix-pipe cycles) beside 340 VALU and 173 scalar instructions: 1 360 cycles of vector issue for 768
// of matrix work, two waves per SIMD -- the loop was bound by its address arithmetic, not by LDS or the matrix pipe
// (ablation, profiles/r05_wgrad3x3_v2.md: without the border masks alone 265 -> 204 us in the measurement build).  What
// the instructions were: the per-lane border masks (8 rows x 2 divisions by invariant divisors per slice), one 32-bit add
// per fragment read (buffer pointer + precomputed offset), a compare + select pair per masked read on top of the bit test,
// the DMA offsets rebuilt from the row index with 64-bit multiplies.  Here:
//   * border flags are computed ONCE per slice row by one wave (64 lanes = 64 rows) when the slice's DMA is issued and
//     left in 64 bytes of LDS beside the slice buffer, laid out so that a lane fetches the flags of its eight rows with one
//     ds_read_b64 a whole slice ahead of their use;
//   * fragment addresses are lane constants + the slice buffer's offset + an immediate (the swizzle is periodic in 16
//     rows, so the four 16-row steps differ by 4 096 bytes): 4 adds per slice for the 16 dY reads; a masked X read is
//     zero-row + flag * (lane constant) -- one bit-field extract and one multiply-add, no compare, no select;
//   * DMA offsets advance by a constant per slice and are range-checked as OFFSETS against lane-constant limits.
// The MFMAs, their order and the LDS images are the kernel's above: the results are bit-identical to it
// (tools/ab_w3g.sh compares the two in the measurement build; tests/test_gpu_h16.py::test_h16_wgrad3x3_kernel holds this one to
// fp64 per tap and to run-to-run bit-identity).
// ---------------------------------------------------------------------------------
constexpr int kW3FlagBase = kW3Lds;                  // kW3NBuf x 64 flag bytes behind the slice buffers
constexpr int kW3LdsV2 = kW3Lds + kW3NBuf * 64;

#ifndef YV4_W3V2_STAGGER
#define YV4_W3V2_STAGGER 1     // build-time A/B (tools/ab_prev.sh): 0 = all eight waves issue DMA(sl + 3) at the same point
#endif
// ABL (measurement build only, compile-time so that the timed kernel carries no extra branches): 1 no DMA inside the loop,
// 2 no workgroup barrier, 4 no MFMAs, 8 no fragment reads, 16 no border masks on the image reads -- wrong results on purpose, to
// time the kernel without a part
template <bool BF16, int ABL = 0>
__global__ __launch_bounds__(kW3Threads, 2) void conv_wgrad3x3_v2_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  typedef __attribute__((address_space(3))) unsigned long long* lds_u64_t;
  typedef __attribute__((address_space(3))) unsigned char* lds_u8_t;
  extern __shared__ __attribute__((aligned(16))) char smem_w3b[];
  constexpr int kRowB = 256;
  constexpr int kDBytes = kW3Rows * kRowB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave >> 2;
  const int wk = wave & 3;

  const int tiles_ci = p.Cin >> 7;
  int tile, chunk;
  if (!wgrad_tile_chunk(p.tiles, p.chunks, p.xcd_map, tile, chunk)) return;
  const int tci = tile % tiles_ci;
  const int kh = (tile / tiles_ci) % 3;
  const int tco = tile / (3 * tiles_ci);
  const int co0 = tco * 128, ci0 = tci * 128;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;
  const int NHW = p.N * p.H * p.W;

  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_w3b;

  // ---- staging (as above): wave w fills rows 8w .. 8w+7 of dY and of the X image, wave 0 also rows 64 .. 67 of the image.
  // Byte offsets of slice 0 and their limits; both advance by a constant per slice.
  const int srow = lane >> 4;
  const int pc = lane & 15;
  auto swz_of = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  const int x_shift = (kh - 1) * p.W - 1;            // image row ir <-> pixel m_slice + ir + x_shift
  const unsigned d_step = (unsigned)(kW3Rows * p.dy_cs * 2), x_step = (unsigned)(kW3Rows * p.x_cs * 2);
  unsigned d_off[2], d_lim[2], x_off[3], x_lim[3];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * wave + 4 * q + srow;
    const int lc = pc ^ swz_of(row);
    const int co = co0 + lc * 8;
    const unsigned cb = (unsigned)((p.dy_co + co) * 2);
    d_off[q] = (unsigned)(m_lo + row) * (unsigned)(p.dy_cs * 2) + cb;
    d_lim[q] = co < p.Cout ? (unsigned)m_hi * (unsigned)(p.dy_cs * 2) + cb : 0u;        // 0: never below -> out of range
    const unsigned xb_ = (unsigned)((p.x_co + ci0 + lc * 8) * 2);
    x_off[q] = (unsigned)(m_lo + row + x_shift) * (unsigned)(p.x_cs * 2) + xb_;          // (a negative pixel wraps to ~2^32)
    x_lim[q] = (unsigned)NHW * (unsigned)(p.x_cs * 2) + xb_;
  }
  {
    const int row = 64 + srow;
    const unsigned xb_ = (unsigned)((p.x_co + ci0 + (pc ^ swz_of(row)) * 8) * 2);
    x_off[2] = (unsigned)(m_lo + row + x_shift) * (unsigned)(p.x_cs * 2) + xb_;
    x_lim[2] = row < 66 ? (unsigned)NHW * (unsigned)(p.x_cs * 2) + xb_ : 0u;             // rows 66, 67 stay zero
  }
  // border flags of slice row r = lane (written by wave 1): position of the byte inside the slice's 64 flag bytes
  const int f_wr = (((lane & 3) * 2 + ((lane >> 3) & 1)) << 3) + ((lane >> 4) << 1) + ((lane >> 2) & 1);
  int f_m = m_lo + lane;                             // (wave 1) the row this lane decodes next
  auto issue = [&](int sl) {
    const int buf = sl & (kW3NBuf - 1);
    const unsigned lb = lds_base + (unsigned)(buf * kW3BufBytes + 8 * wave * kRowB);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      lds_dma16_t(rsD, lb + (unsigned)(4 * q * kRowB), d_off[q] < d_lim[q] ? d_off[q] : kOOB, 0u);
      lds_dma16_t(rsX, lb + (unsigned)(kDBytes + 4 * q * kRowB), x_off[q] < x_lim[q] ? x_off[q] : kOOB, 0u);
      d_off[q] += d_step;
      x_off[q] += x_step;
    }
    if (wave == 0) {
      lds_dma16_t(rsX, lds_base + (unsigned)(buf * kW3BufBytes + kDBytes + 64 * kRowB), x_off[2] < x_lim[2] ? x_off[2] : kOOB, 0u);
      x_off[2] += x_step;
    }
    if (wave == 1) {
      unsigned b3 = 0u;
      if (f_m < m_hi) {
        const int n = fd_div(f_m, p.fd_hw);
        const int rm = f_m - n * (p.H * p.W);
        const int ho = fd_div(rm, p.fd_wo);
        const int wo = rm - ho * p.W;
        if ((unsigned)(ho + kh - 1) < (unsigned)p.H) b3 = (wo > 0 ? 1u : 0u) | 2u | (wo + 1 < p.W ? 4u : 0u);
      }
      *(lds_u8_t)(smem_w3b + kW3FlagBase + buf * 64 + f_wr) = (unsigned char)b3;
      f_m += kW3Rows;
    }
  };

  // ---- transposed fragment reads: lane = 16 g + 4 qq + pp supplies row (block + qq), columns 4 pp .. 4 pp + 3 of its
  // 16-column half; step S adds 16 rows = 4 096 bytes (the swizzle only sees the row's low four bits)
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  auto row_addr = [&](int row, int col_base) -> int {
    const int chunk_ = (col_base + 16 * colhalf) / 8 + (pp >> 1);
    return kRowB * row + 16 * (chunk_ ^ swz_of(row)) + 8 * (pp & 1);
  };
  constexpr int kZeroRd = kDBytes + kW3ZeroRow * kRowB;
  int a_base[2][2];                                  // dY: [co tile a][j], step 0, buffer 0
  int b_dlt[2][3][4];                                // X: (address of row 16 S + 8 hh + 4 j + qq + kw) - (zero row), [j][kw][S]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int a = 0; a < 2; ++a) a_base[a][j] = row_addr(8 * hh + 4 * j + qq, wc * 64 + a * 32);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int S = 0; S < 4; ++S) b_dlt[j][kw][S] = kDBytes + row_addr(16 * S + 8 * hh + 4 * j + qq + kw, wk * 32) - kZeroRd;
  }
  const int f_rd = kW3FlagBase + ((qq * 2 + hh) << 3);

  f32x16 acc[2][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  const int nsl = (m_hi - m_lo + kW3Rows - 1) / kW3Rows;
  s16x8_t fa[2][2] = {}, fb[2][3] = {};              // fragment sets: step s computes from set s & 1
  // FL: the eight flag bytes of this lane's rows of the slice ([S][j], bits kw); BO: the slice buffer's byte offset
#define YV4_W3B_LOAD(SET, BO, S, FL)                                                                          \
  if constexpr (!(ABL & 8)) {                                                                                 \
    const char* ab_ = smem_w3b + (BO) + 4096 * (S);                                                           \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                           \
      const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(ab_ + a_base[a][0]));            \
      const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(ab_ + a_base[a][1]));            \
      fa[SET][a] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);                                   \
    }                                                                                                         \
    const unsigned fw_ = (unsigned)((FL) >> (((S) >> 1) * 32));                                               \
    const int zb_ = (BO) + kZeroRd;                                                                           \
    _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                                        \
      const int f0_ = (ABL & 16) ? 1 : (int)((fw_ >> ((((S) & 1) * 2 + 0) * 8 + kw)) & 1u);                   \
      const int f1_ = (ABL & 16) ? 1 : (int)((fw_ >> ((((S) & 1) * 2 + 1) * 8 + kw)) & 1u);                   \
      const int r0_ = __mul24(f0_, b_dlt[0][kw][S]) + zb_;                                   \
      const int r1_ = __mul24(f1_, b_dlt[1][kw][S]) + zb_;                                   \
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(smem_w3b + r0_));                 \
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(smem_w3b + r1_));                 \
      fb[SET][kw] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);                                  \
    }                                                                                                         \
  }
#define YV4_W3B_MFMA(SET)                                                                                     \
  {                                                                                                           \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                             \
      _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) {                                                      \
        if constexpr (ABL & 4) { asm volatile("" :: "v"(fa[SET][a]), "v"(fb[SET][kw])); continue; }          \
        if (BF16)                                                                                             \
          acc[a][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET][a]),      \
                                                               __builtin_bit_cast(bf16x8_w, fb[SET][kw]), acc[a][kw], 0, 0, 0); \
        else                                                                                                  \
          acc[a][kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET][a]),        \
                                                              __builtin_bit_cast(f16x8_w, fb[SET][kw]), acc[a][kw], 0, 0, 0); \
      }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
  // pieces per slice: 4 (5 on wave 0); DMA(t + 2) and DMA(t + 3) may stay in flight at the wait of slice t
#define YV4_W3B_WAIT()                                                                                        \
  {                                                                                                           \
    if (wave == 0) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");                               \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                          \
  }

  issue(0);
  issue(1);
  issue(2);
  YV4_W3B_WAIT();                                     // DMA(0) landed (newer: 1, 2); the flag bytes are written
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned long long fl = *(lds_u64_t)(smem_w3b + f_rd);
  YV4_W3B_LOAD(0, 0, 0, fl);
  for (int sl = 0; sl < nsl; ++sl) {
    const int bo = (sl & (kW3NBuf - 1)) * kW3BufBytes;
    const int nb = (sl + 1) & (kW3NBuf - 1);
    const int nbo = nb * kW3BufBytes;
    // flags of slice sl + 1: written when its DMA was issued (two barriers ago), wanted after this slice's barrier
    const unsigned long long fln = *(lds_u64_t)(smem_w3b + f_rd + nb * 64);
    YV4_W3B_LOAD(1, bo, 1, fl);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_MFMA(0);
    // DMA(sl + 3) into the buffer slice sl - 1 read (freed by the previous barrier): waves 0-3 issue their 4-5 pieces
    // here, waves 4-7 (their partners on the SIMDs) one MFMA step later -- issued by all eight waves at the same point
    // the pieces' 400-500 issue cycles left the matrix pipe idle (compile-time ablation: -10 % without the DMA)
    if constexpr (!(ABL & 1)) { if (wave < 4 || !YV4_W3V2_STAGGER) issue(sl + 3); }
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_LOAD(0, bo, 2, fl);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_MFMA(1);
    if constexpr (!(ABL & 1)) { if (wave >= 4 && YV4_W3V2_STAGGER) issue(sl + 3); }
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_LOAD(1, bo, 3, fl);
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_MFMA(0);
    if constexpr (!(ABL & 1)) YV4_W3B_WAIT()          // own DMA(sl + 1) landed; every read of slice sl has returned
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(ABL & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    YV4_W3B_LOAD(0, nbo, 0, fln);                     // (beyond the last slice: zero-filled buffers, never used)
    __builtin_amdgcn_sched_barrier(0);
    YV4_W3B_MFMA(1);
    fl = fln;
  }
#undef YV4_W3B_WAIT
#undef YV4_W3B_MFMA
#undef YV4_W3B_LOAD
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the dummy tail DMAs must land before the LDS is released

  // D[row = co][col = ci]: row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31
  const int r = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int kcol = (kh * 3 + kw) * p.Cin + ci0 + wk * 32 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wc * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
        if (co < p.Cout) {
          if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[a][kw][e];
          else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[a][kw][e]);
        }
      }
    }
}

// ---------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride-1 / pad-1 layers with FEW channels (Cin 16, 32 or 64 per pixel, Cout 32 or 64):
// the stem, the first Bottleneck and the first CSP stage of CSPDarknet at 608 / 304 / 152 pixels.  These layers are bound
// by their bytes (dY + X once = 0.23-0.45 ms at batch 64) and the 128 x 128 tiles above serve them badly: dW is 32-64
// rows by 72-576 columns, so a tile is mostly padding, every column tile re-reads dY, and the im2col operand fetches
// every source pixel nine times -- 960 bytes of LDS fill per output pixel for the 32 -> 64 layer, three times what the
// memory system delivers per unit time to 64 KB of slice buffers per CU (0.65-0.9 ms per layer).
//
// Here an 8-wave workgroup (two per CU) owns ALL of dW and a chunk of the M reduction (flattened pixel index m, as in the
// kernel above).  A 64-row slice is the rows of dY plus THREE images of the source pixels, one per kh (image row ir <->
// pixel m + ir + (kh - 1) W - 1, 66 rows; the three kw taps of a kh read one image, operand row = reduction row + kw):
// 64 Cout * 2 + 3 * 66 * Cin * 2 bytes -- 323 per output pixel for 32 -> 64.  Rows keep their natural pitch (32 / 64 /
// 128 bytes); 128-byte rows swap their 64-byte halves on rows 2, 3 (mod 4) so that the four rows of a transposed block
// fall into different banks.  Wave w < 3 * (Cout / 32) computes the (kh = w % 3, 32-row co block w / 3) part of dW: its
// dY fragment is shared by its 2-6 column blocks (kw x channel halves); all eight waves fill the buffers.  Borders are
// masked per lane of the transposed reads (a lane supplies ONE reduction row: redirected to the image's zero row it
// contributes nothing).  Two to four slice buffers per workgroup (two workgroups per CU), one barrier per slice, every wave issues
// the same number of DMA instructions per slice (dummies into a scratch KB) so that one counted wait serves all.
// Same chunked, deterministic output as the kernels above.
// ---------------------------------------------------------------------------------
constexpr int kFcThreads = 512;
constexpr int kFcRows = 64;
constexpr int kFcZeroRow = 66;

template <int CIN, int COUT> struct FcGeom {
  static constexpr int PX = CIN * 2;                                   // bytes per pixel of an X image row
  static constexpr int CPP = PX / 16;                                  // 16-byte chunks per pixel
  static constexpr int XRows = CIN == 16 ? 96 : (CIN == 32 ? 80 : 72); // >= 68 and XRows * CPP % 64 == 0
  static constexpr int XPieces = XRows * CPP / 64;                     // DMA instructions per image
  static constexpr int XBytes = XRows * PX;
  static constexpr int DP = COUT * 2;                                  // bytes per dY row
  static constexpr int CPD = DP / 16;
  static constexpr int DPieces = kFcRows * CPD / 64;
  static constexpr int DBytes = kFcRows * DP;
  static constexpr int NBK = CIN == 16 ? 2 : CIN / 32 * 3;             // 32-column blocks of dW per kh
  static constexpr int CB = COUT / 32;
  static constexpr int BufBytes = DBytes + 3 * XBytes;
  // slice buffers of ONE workgroup.  Cin 16 / 32: two workgroups share a CU (12 computing waves = three per SIMD: with
  // one workgroup the six computing waves sit two-two-one-one on the SIMDs and the pair sets the pace).  Cin 64: six
  // accumulators and two fragment sets are 234 VGPRs -- one workgroup per CU, four buffers.
  // (One workgroup with eleven buffers for the narrowest geometry -- Cin 16, Cout 32, 13 KB per slice, whose rate is
  // (bytes in flight) / latency: 62 KB per CU = 17 GB/s per CU -- was measured: 883 -> 1 286 us.  With three computing
  // waves per CU nothing hides a slice's barrier -> masks -> reads -> MFMA chain, ~0.9 us per 64 rows.)
  static constexpr int WGs = CIN == 64 ? 1 : 2;
  static constexpr int NBuf = WGs == 1 ? 4 : (BufBytes * 4 + 1024 <= 80 * 1024 ? 4 : (BufBytes * 3 + 1024 <= 80 * 1024 ? 3 : 2));
  static constexpr int Pieces = DPieces + 3 * XPieces;
  static constexpr int PW = (Pieces + 7) / 8;                          // DMA instructions per wave and slice
  static constexpr int Lds = NBuf * BufBytes + 1024;                   // + the dummies' scratch
};

#ifdef YV4_MEASURE   // the FIRST form of the few-channel weight gradient: the measurement build's A/B partner of the second form (same bits); the product takes the generic 16-bit kernel where the second form does not apply
template <bool BF16, int CIN, int COUT>
__global__ __launch_bounds__(kFcThreads, (FcGeom<CIN, COUT>::WGs)) void conv_wgrad_fc_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef FcGeom<CIN, COUT> G;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  extern __shared__ __attribute__((aligned(16))) char smem_fc[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = (int)blockIdx.x;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;
  const int NHW = p.N * p.H * p.W;
  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_fc;
  auto swz128 = [](int row) { return ((row >> 1) & 1) << 2; };         // 128-byte rows only

  // ---- staging: piece t = wave + 8 i of a slice.  Per lane and piece, fixed for the kernel: the row inside the slice
  // (a huge one where the lane never fetches: padding rows of an image, dY columns past Cout) and the byte offset of its
  // 16 bytes at slice 0; per slice the offset advances by 64 rows -- the VALU work of an issue is an add, a range
  // check and a select per piece (the first version decoded rows and columns per slice and the kernel was bound by its
  // 132 M VALU instructions, not by LDS or HBM).
  constexpr int kNever = 1 << 28;
  int pc_kind[G::PW], pc_row[G::PW];                 // kind: 0 dY, 1..3 image kh = kind - 1, -1 dummy (wave-uniform)
  unsigned pc_off0[G::PW], pc_lds[G::PW];
#pragma unroll
  for (int i = 0; i < G::PW; ++i) {
    const int t = wave + 8 * i;
    pc_kind[i] = -1; pc_row[i] = kNever; pc_off0[i] = 0u; pc_lds[i] = 0u;
    if (t < G::DPieces) {
      const int L = 64 * t + lane;
      const int row = L / G::CPD, pc = L % G::CPD;
      const int lc = G::CPD == 8 ? (pc ^ swz128(row)) : pc;
      pc_kind[i] = 0;
      pc_row[i] = lc * 8 < p.Cout ? row : kNever;
      pc_off0[i] = (unsigned)((((int64_t)(m_lo + row) * p.dy_cs) + p.dy_co + lc * 8) * 2);
      pc_lds[i] = (unsigned)(t * 1024);
    } else if (t < G::Pieces) {
      const int u = t - G::DPieces;
      const int khp = u / G::XPieces, q = u - khp * G::XPieces;
      const int L = 64 * q + lane;
      const int row = L / G::CPP, pc = L % G::CPP;
      const int lc = G::CPP == 8 ? (pc ^ swz128(row)) : pc;
      pc_kind[i] = 1 + khp;
      pc_row[i] = row < kFcZeroRow ? row + (khp - 1) * p.W - 1 : kNever;      // pixel = m_base + pc_row
      pc_off0[i] = (unsigned)((((int64_t)(m_lo + row + (khp - 1) * p.W - 1) * p.x_cs) + p.x_co + lc * 8) * 2);
      pc_lds[i] = (unsigned)(G::DBytes + khp * G::XBytes + q * 1024);
    }
  }
  const unsigned d_step = (unsigned)(kFcRows * p.dy_cs * 2), x_step = (unsigned)(kFcRows * p.x_cs * 2);
  const int HW = p.H * p.W;
  auto issue = [&](int sl, int nsl) {
    const int buf = sl % G::NBuf;
    const int m_base = m_lo + sl * kFcRows;
    const bool live = sl < nsl;
    // Vertical borders (uniform per slice): a source pixel of an image's LAST row can only be "the row above" of the next
    // image's first row, one of its FIRST row only "the row below" of the previous image's last -- such pixels must
    // arrive as zeros (the horizontal border is the readers' per-row mask).  Whether the 66 pixels of the kh = 0 / 2
    // images touch such a row is a property of the slice; only then do the lanes look at their own pixel's row.
    bool edge[3] = {false, false, false};
#pragma unroll
    for (int k = 0; k < 3; k += 2) {
      int a0 = m_base + (k - 1) * p.W - 1, a1 = a0 + kFcZeroRow - 1;
      a0 = a0 < 0 ? 0 : a0;
      a1 = a1 >= NHW ? NHW - 1 : a1;
      if (a0 <= a1) {
        const int r0 = fd_div(a0, p.fd_wo), r1 = fd_div(a1, p.fd_wo);
        const int n0 = fd_div(a0, p.fd_hw);
        const int h0 = r0 - n0 * p.H;                                // row of the first pixel inside its image
        edge[k] = k == 0 ? (h0 + (r1 - r0) >= p.H - 1) : (h0 == 0 || h0 + (r1 - r0) >= p.H);
      }
    }
#pragma unroll
    for (int i = 0; i < G::PW; ++i) {
      unsigned off = kOOB;
      unsigned dst = lds_base + (unsigned)(G::NBuf * G::BufBytes);       // dummy: the scratch KB
      if (pc_kind[i] == 0) {
        if (live && m_base + pc_row[i] < m_hi) off = pc_off0[i] + (unsigned)sl * d_step;
        dst = lds_base + (unsigned)(buf * G::BufBytes) + pc_lds[i];
        lds_dma16_t(rsD, dst, off, 0u);
      } else {
        if (pc_kind[i] > 0) {
          const int pix = m_base + pc_row[i];
          bool ok = live && (unsigned)pix < (unsigned)NHW;
          if (pc_kind[i] != 2 && edge[pc_kind[i] - 1]) {
            const int n = fd_div(pix, p.fd_hw);
            const int hs = fd_div(pix - n * HW, p.fd_wo);
            ok = ok && hs != (pc_kind[i] == 1 ? p.H - 1 : 0);
          }
          if (ok) off = pc_off0[i] + (unsigned)sl * x_step;
          dst = lds_base + (unsigned)(buf * G::BufBytes) + pc_lds[i];
        }
        lds_dma16_t(rsX, dst, off, 0u);
      }
    }
  };

  // ---- compute roles
  // measurement-only bits (YV4_WFC_ABLATE): 1 no fragment reads / MFMAs, 2 no DMA, 4 no border masks, 8 no MFMAs
  // Which waves compute.  A wave sits on SIMD (wave mod 4) and a computing wave keeps its SIMD busy for most of a slice
  // (its VALU instructions take four cycles each and its MFMAs queue behind one another), so six roles on waves 0..5
  // load the SIMDs 2-2-1-1 and the pair sets the pace of every slice.  The two workgroups that share a CU (b and b + 256
  // of a one-round grid) therefore start their roles two waves apart: together 3-3-3-3.
  const int role = (wave + 8 - 2 * (((int)blockIdx.x >> 8) & 1)) & 7;
  const bool computes = role < 3 * G::CB && !YV4_ABLATE(p.ablate, 1);
  const int kh = role % 3, cb = role / 3;
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  // dY fragment addresses (inside a buffer): [step s][j]
  unsigned d_rd[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * s + 8 * hh + 4 * j + qq;
      const int chunk16 = (cb * 32 + 16 * colhalf) / 8 + (pp >> 1);
      d_rd[s][j] = (unsigned)(row * G::DP + ((G::CPD == 8 ? (chunk16 ^ swz128(row)) : chunk16) << 4) + 8 * (pp & 1));
    }
  // X fragment addresses (inside a buffer) of (step s, j, column block nb), fixed for the kernel: the lane's 16-column
  // half decides tap and channel base; x_zr[nb] = the same columns of the image's zero row
  const unsigned ximg = (unsigned)(G::DBytes + kh * G::XBytes);
  auto kw_of = [&](int nb) -> int { return CIN == 16 ? 2 * nb + colhalf : (CIN == 32 ? nb : nb >> 1); };
  auto x_addr = [&](int row, int nb) -> unsigned {
    const int kw = kw_of(nb);
    const int cib = CIN == 16 ? 0 : (CIN == 32 ? 16 * colhalf : 32 * (nb & 1) + 16 * colhalf);
    const int r = kw < 3 ? row + kw : kFcZeroRow;
    const int chunk16 = cib / 8 + (pp >> 1);
    return ximg + (unsigned)(r * G::PX + ((G::CPP == 8 ? (chunk16 ^ swz128(r)) : chunk16) << 4) + 8 * (pp & 1));
  };
  unsigned x_rd[4][2][G::NBK], x_zr[G::NBK];
#pragma unroll
  for (int nb = 0; nb < G::NBK; ++nb) {
    x_zr[nb] = x_addr(kFcZeroRow - kw_of(nb) < 0 ? 0 : kFcZeroRow - (kw_of(nb) < 3 ? kw_of(nb) : 0), nb);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) x_rd[s][j][nb] = x_addr(16 * s + 8 * hh + 4 * j + qq, nb);
  }

  f32x16 acc[G::NBK];
#pragma unroll
  for (int b = 0; b < G::NBK; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

  const int nsl = (m_hi - m_lo + kFcRows - 1) / kFcRows;
  // horizontal border masks of this lane's eight reduction rows of a slice: bit (s * 2 + j) * 3 + kw set = the row's
  // column wo + kw - 1 exists.  (Rows past the end of the chunk need no mask: their dY rows arrive as zeros.)  No
  // branches: the eight rows of every lane are computed alike.
  auto slice_masks = [&](int sl) -> unsigned {
    unsigned mk = 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = m_lo + sl * kFcRows + 16 * s + 8 * hh + 4 * j + qq;
        const int wo = m - fd_div(m, p.fd_wo) * p.W;
        const unsigned b3 = (wo > 0 ? 1u : 0u) | 2u | (wo + 1 < p.W ? 4u : 0u);
        mk |= b3 << ((s * 2 + j) * 3);
      }
    return mk;
  };

#pragma unroll
  for (int s0 = 0; s0 < G::NBuf - 1; ++s0) issue(s0, nsl);
  for (int sl = 0; sl < nsl; ++sl) {
    // own DMA(sl) landed: the NBuf - 2 younger slices (PW instructions each) may stay in flight
    constexpr int kLeft = (G::NBuf - 2) * G::PW;
    static_assert(kLeft >= 0 && kLeft < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kLeft) : "memory");
    __builtin_amdgcn_s_barrier();                      // slice sl complete in LDS; every wave is done with slice sl - 1
    asm volatile("" ::: "memory");
    if (!YV4_ABLATE(p.ablate, 2)) issue(sl + G::NBuf - 1, nsl);       // into the buffer slice sl - 1 left
    if (computes) {
      const char* bufp = smem_fc + (sl % G::NBuf) * G::BufBytes;
      // horizontal borders: a slice whose 64 pixels lie inside one image row, away from its ends, needs no mask (uniform)
      unsigned mk = 0xFFFFFFu;
      {
        const int mb = m_lo + sl * kFcRows;
        const int rb = fd_div(mb, p.fd_wo);
        const int wb = mb - rb * p.W;
        if (!(wb > 0 && wb + kFcRows < p.W) && !YV4_ABLATE(p.ablate, 4)) mk = slice_masks(sl);
      }
      // two fragment sets: the reads of step s + 1 are issued in front of the MFMAs of step s (left to itself the
      // compiler reuses one register set and every MFMA waits out a fresh LDS round trip: 2 400 cycles per slice)
      s16x8_t fa[2], fb[2][G::NBK];
#define YV4_FC_LOAD(SET, S)                                                                                   \
      {                                                                                                       \
        const s16x4_t a0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + d_rd[S][0]));          \
        const s16x4_t a1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + d_rd[S][1]));          \
        fa[SET] = __builtin_shufflevector(a0_, a1_, 0, 1, 2, 3, 4, 5, 6, 7);                                  \
        _Pragma("unroll") for (int nb = 0; nb < G::NBK; ++nb) {                                               \
          const int kw_ = kw_of(nb);                                                                          \
          const int kb_ = kw_ < 3 ? kw_ : 0;                                                                  \
          const bool ok0_ = kw_ < 3 && ((mk >> (((S) * 2 + 0) * 3 + kb_)) & 1u);                              \
          const bool ok1_ = kw_ < 3 && ((mk >> (((S) * 2 + 1) * 3 + kb_)) & 1u);                              \
          const s16x4_t b0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + (ok0_ ? x_rd[S][0][nb] : x_zr[nb]))); \
          const s16x4_t b1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + (ok1_ ? x_rd[S][1][nb] : x_zr[nb]))); \
          fb[SET][nb] = __builtin_shufflevector(b0_, b1_, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        }                                                                                                     \
      }
#define YV4_FC_MFMA(SET)                                                                                      \
      {                                                                                                       \
        _Pragma("unroll") for (int nb = 0; nb < G::NBK; ++nb) {                                               \
          if (YV4_ABLATE(p.ablate, 8)) { acc[nb][0] += __builtin_bit_cast(float, (int)(fa[SET][0] + fb[SET][nb][0])); continue; } \
          if (BF16)                                                                                           \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET]),          \
                                                              __builtin_bit_cast(bf16x8_w, fb[SET][nb]), acc[nb], 0, 0, 0); \
          else                                                                                                \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET]),            \
                                                             __builtin_bit_cast(f16x8_w, fb[SET][nb]), acc[nb], 0, 0, 0);   \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }
      YV4_FC_LOAD(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_LOAD(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(0);
      YV4_FC_LOAD(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(1);
      YV4_FC_LOAD(1, 3);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(0);
      YV4_FC_MFMA(1);
#undef YV4_FC_MFMA
#undef YV4_FC_LOAD
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the tail's out-of-range DMAs must land before the LDS goes

  if (!computes) return;
  // D[row = co][col]: row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31 -> (kw, ci) of the block
  const int ncol = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < G::NBK; ++nb) {
    int kw, ci;
    if (CIN == 16) { kw = 2 * nb + (ncol >> 4); ci = ncol & 15; }
    else if (CIN == 32) { kw = nb; ci = ncol; }
    else { kw = nb >> 1; ci = 32 * (nb & 1) + ncol; }
    if (kw >= 3 || ci >= p.Cin) continue;
    const int kcol = (kh * 3 + kw) * p.Cin + ci;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
      if (co < p.Cout) {
        if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[nb][e];
        else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[nb][e]);
      }
    }
  }
}
#endif  // YV4_MEASURE

// ---------------------------------------------------------------------------------
// The few-channel kernel with its per-slice bookkeeping taken off the lanes (round 5; see conv_wgrad3x3_v2_h16_kernel).  The
// first form issues, per 64-row slice and wave, 8-24 MFMAs beside 213-316 VALU and 219-306 scalar instructions: its ~1.2 us
// per slice on the stem's geometry IS that instruction stream (8 MFMAs = 256 matrix cycles against ~2 000 issue cycles).
// Here the border flags are computed once per slice row by the idle role-7 wave (vertical borders included, so the DMA no
// longer zeroes anything and the two scalar row decodes per slice are gone), a lane fetches its eight rows' flags with one
// ds_read_b64 a slice ahead, and the DMA offsets advance by constants and are range-checked as offsets.  Same images, same
// fragment addresses, same MFMAs in the same order: dW is bit-identical to the first form (tools/ab_wfc.sh).
// ---------------------------------------------------------------------------------
template <bool BF16, int CIN, int COUT>
__global__ __launch_bounds__(kFcThreads, (FcGeom<CIN, COUT>::WGs)) void conv_wgrad_fc_v2_h16_kernel(WgradArgs p, unsigned x_bytes, unsigned dy_bytes) {
  typedef FcGeom<CIN, COUT> G;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4_t;
  typedef __attribute__((address_space(3))) unsigned long long* lds_u64_t;
  typedef __attribute__((address_space(3))) unsigned char* lds_u8_t;
  extern __shared__ __attribute__((aligned(16))) char smem_fc[];
  constexpr int kFlagBase = G::Lds;                  // NBuf x 3 (kh) x 64 flag bytes behind the buffers and the scratch KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = (int)blockIdx.x;
  const int m_lo = chunk * p.rows_per_chunk;
  const int m_hi = min(m_lo + p.rows_per_chunk, p.M);
  if (m_lo >= m_hi) return;
  const int NHW = p.N * p.H * p.W;
  const u32x4_t rsX = make_rsrc_t(p.x, x_bytes);
  const u32x4_t rsD = make_rsrc_t(p.dy, dy_bytes);
  constexpr unsigned kOOB = 0xFFFFFFF0u;
  const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)smem_fc;
  auto swz128 = [](int row) { return ((row >> 1) & 1) << 2; };         // 128-byte rows only

  // ---- staging: piece t = wave + 8 i of a slice.  Per lane and piece, fixed for the kernel: the row inside the slice
  // (a huge one where the lane never fetches: padding rows of an image, dY columns past Cout) and the byte offset of its
  // 16 bytes at slice 0; per slice the offset advances by 64 rows -- the VALU work of an issue is an add, a range
  // check and a select per piece (the first version decoded rows and columns per slice and the kernel was bound by its
  // 132 M VALU instructions, not by LDS or HBM).
  int pc_kind[G::PW];                                // kind: 0 dY, 1..3 image kh = kind - 1, -1 dummy (wave-uniform)
  unsigned pc_off[G::PW], pc_lim[G::PW], pc_lds[G::PW];   // byte offset of the next slice's 16 bytes, its limit (0: never)
#pragma unroll
  for (int i = 0; i < G::PW; ++i) {
    const int t = wave + 8 * i;
    pc_kind[i] = -1; pc_off[i] = 0u; pc_lim[i] = 0u; pc_lds[i] = 0u;
    if (t < G::DPieces) {
      const int L = 64 * t + lane;
      const int row = L / G::CPD, pc = L % G::CPD;
      const int lc = G::CPD == 8 ? (pc ^ swz128(row)) : pc;
      pc_kind[i] = 0;
      const unsigned cb_ = (unsigned)((p.dy_co + lc * 8) * 2);
      pc_off[i] = (unsigned)(m_lo + row) * (unsigned)(p.dy_cs * 2) + cb_;
      pc_lim[i] = lc * 8 < p.Cout ? (unsigned)m_hi * (unsigned)(p.dy_cs * 2) + cb_ : 0u;
      pc_lds[i] = (unsigned)(t * 1024);
    } else if (t < G::Pieces) {
      const int u = t - G::DPieces;
      const int khp = u / G::XPieces, q = u - khp * G::XPieces;
      const int L = 64 * q + lane;
      const int row = L / G::CPP, pc = L % G::CPP;
      const int lc = G::CPP == 8 ? (pc ^ swz128(row)) : pc;
      pc_kind[i] = 1 + khp;
      const unsigned xb_ = (unsigned)((p.x_co + lc * 8) * 2);
      // pixel = m_base + row + (khp - 1) W - 1; a pixel in front of the map wraps to ~2^32 and fails the limit
      pc_off[i] = (unsigned)(m_lo + row + (khp - 1) * p.W - 1) * (unsigned)(p.x_cs * 2) + xb_;
      pc_lim[i] = row < kFcZeroRow ? (unsigned)NHW * (unsigned)(p.x_cs * 2) + xb_ : 0u;
      pc_lds[i] = (unsigned)(G::DBytes + khp * G::XBytes + q * 1024);
    }
  }
  const unsigned d_step = (unsigned)(kFcRows * p.dy_cs * 2), x_step = (unsigned)(kFcRows * p.x_cs * 2);
  // Borders are the READERS' business here: a source pixel that lies in another image row / image than the tap wants is
  // fetched like any other and masked per lane through the flags below (the first form zeroed the vertical ones at DMA
  // time, which cost every slice two scalar row decodes and, on border slices, a decode per lane and piece).
  // Flags of slice row r = lane, one byte per kh (bits kw), written by the wave of role 7 (it never computes) when the
  // slice's DMA is issued; byte position inside the 64 of a (buffer, kh): ((qq * 2 + hh) << 3) + s * 2 + j for
  // r = 16 s + 8 hh + 4 j + qq, so that a lane fetches its eight rows' flags with one ds_read_b64.
  const int f_wr = (((lane & 3) * 2 + ((lane >> 3) & 1)) << 3) + ((lane >> 4) << 1) + ((lane >> 2) & 1);
  int f_m = m_lo + lane;
  const int role_w = (wave + 8 - 2 * (((int)blockIdx.x >> 8) & 1)) & 7;
  auto issue = [&](int sl) {
    const int buf = sl % G::NBuf;
#pragma unroll
    for (int i = 0; i < G::PW; ++i) {
      const unsigned off = pc_off[i] < pc_lim[i] ? pc_off[i] : kOOB;
      if (pc_kind[i] == 0) {
        lds_dma16_t(rsD, lds_base + (unsigned)(buf * G::BufBytes) + pc_lds[i], off, 0u);
        pc_off[i] += d_step;
      } else if (pc_kind[i] > 0) {
        lds_dma16_t(rsX, lds_base + (unsigned)(buf * G::BufBytes) + pc_lds[i], off, 0u);
        pc_off[i] += x_step;
      } else {
        lds_dma16_t(rsX, lds_base + (unsigned)(G::NBuf * G::BufBytes), kOOB, 0u);     // dummy: the scratch KB
      }
    }
    if (role_w == 7) {
      unsigned b0 = 0u, b1 = 0u, b2 = 0u;
      if (f_m < m_hi) {
        const int n = fd_div(f_m, p.fd_hw);
        const int rm = f_m - n * (p.H * p.W);
        const int ho = fd_div(rm, p.fd_wo);
        const int wo = rm - ho * p.W;
        const unsigned h3 = (wo > 0 ? 1u : 0u) | 2u | (wo + 1 < p.W ? 4u : 0u);
        b0 = ho > 0 ? h3 : 0u;
        b1 = h3;
        b2 = ho + 1 < p.H ? h3 : 0u;
      }
      lds_u8_t fp = (lds_u8_t)(smem_fc + kFlagBase + buf * 192 + f_wr);
      fp[0] = (unsigned char)b0; fp[64] = (unsigned char)b1; fp[128] = (unsigned char)b2;
      f_m += kFcRows;
    }
  };

  // ---- compute roles
  // measurement-only bits (YV4_WFC_ABLATE): 1 no fragment reads / MFMAs, 2 no DMA, 4 no border masks, 8 no MFMAs
  // Which waves compute.  A wave sits on SIMD (wave mod 4) and a computing wave keeps its SIMD busy for most of a slice
  // (its VALU instructions take four cycles each and its MFMAs queue behind one another), so six roles on waves 0..5
  // load the SIMDs 2-2-1-1 and the pair sets the pace of every slice.  The two workgroups that share a CU (b and b + 256
  // of a one-round grid) therefore start their roles two waves apart: together 3-3-3-3.
  const int role = role_w;
  const bool computes = role < 3 * G::CB;
  const int kh = role % 3, cb = role / 3;
  const int g = lane >> 4, i16 = lane & 15;
  const int hh = g >> 1, colhalf = g & 1;
  const int qq = i16 >> 2, pp = i16 & 3;
  // dY fragment addresses (inside a buffer): [step s][j]
  unsigned d_rd[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * s + 8 * hh + 4 * j + qq;
      const int chunk16 = (cb * 32 + 16 * colhalf) / 8 + (pp >> 1);
      d_rd[s][j] = (unsigned)(row * G::DP + ((G::CPD == 8 ? (chunk16 ^ swz128(row)) : chunk16) << 4) + 8 * (pp & 1));
    }
  // X fragment addresses (inside a buffer) of (step s, j, column block nb), fixed for the kernel: the lane's 16-column
  // half decides tap and channel base; x_zr[nb] = the same columns of the image's zero row
  const unsigned ximg = (unsigned)(G::DBytes + kh * G::XBytes);
  auto kw_of = [&](int nb) -> int { return CIN == 16 ? 2 * nb + colhalf : (CIN == 32 ? nb : nb >> 1); };
  auto x_addr = [&](int row, int nb) -> unsigned {
    const int kw = kw_of(nb);
    const int cib = CIN == 16 ? 0 : (CIN == 32 ? 16 * colhalf : 32 * (nb & 1) + 16 * colhalf);
    const int r = kw < 3 ? row + kw : kFcZeroRow;
    const int chunk16 = cib / 8 + (pp >> 1);
    return ximg + (unsigned)(r * G::PX + ((G::CPP == 8 ? (chunk16 ^ swz128(r)) : chunk16) << 4) + 8 * (pp & 1));
  };
  unsigned x_rd[4][2][G::NBK], x_zr[G::NBK];
#pragma unroll
  for (int nb = 0; nb < G::NBK; ++nb) {
    x_zr[nb] = x_addr(kFcZeroRow - kw_of(nb) < 0 ? 0 : kFcZeroRow - (kw_of(nb) < 3 ? kw_of(nb) : 0), nb);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j) x_rd[s][j][nb] = x_addr(16 * s + 8 * hh + 4 * j + qq, nb);
  }

  f32x16 acc[G::NBK];
#pragma unroll
  for (int b = 0; b < G::NBK; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

  const int nsl = (m_hi - m_lo + kFcRows - 1) / kFcRows;
  const int f_rd = kFlagBase + kh * 64 + ((qq * 2 + hh) << 3);
  static_assert(G::NBuf >= 3, "the flags of slice sl + 1 are read during slice sl: they must have been written a barrier ago");
#pragma unroll
  for (int s0 = 0; s0 < G::NBuf - 1; ++s0) issue(s0);
  unsigned long long fl = 0ull, fln = 0ull;
  for (int sl = 0; sl < nsl; ++sl) {
    // own DMA(sl) landed: the NBuf - 2 younger slices (PW instructions each) may stay in flight
    constexpr int kLeft = (G::NBuf - 2) * G::PW;
    static_assert(kLeft >= 0 && kLeft < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kLeft) : "memory");
    __builtin_amdgcn_s_barrier();                      // slice sl complete in LDS; every wave is done with slice sl - 1
    asm volatile("" ::: "memory");
    issue(sl + G::NBuf - 1);                           // into the buffer slice sl - 1 left
    if (computes) {
      const char* bufp = smem_fc + (sl % G::NBuf) * G::BufBytes;
      // this slice's flags (slice 0: fetched now; later slices: fetched one slice ahead) and the next slice's
      if (sl == 0) fl = *(lds_u64_t)(smem_fc + f_rd);
      fln = *(lds_u64_t)(smem_fc + f_rd + ((sl + 1) % G::NBuf) * 192);
      // two fragment sets: the reads of step s + 1 are issued in front of the MFMAs of step s (left to itself the
      // compiler reuses one register set and every MFMA waits out a fresh LDS round trip: 2 400 cycles per slice)
      s16x8_t fa[2], fb[2][G::NBK];
#define YV4_FC_LOAD(SET, S)                                                                                   \
      {                                                                                                       \
        const s16x4_t a0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + d_rd[S][0]));          \
        const s16x4_t a1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + d_rd[S][1]));          \
        fa[SET] = __builtin_shufflevector(a0_, a1_, 0, 1, 2, 3, 4, 5, 6, 7);                                  \
        _Pragma("unroll") for (int nb = 0; nb < G::NBK; ++nb) {                                               \
          const int kw_ = kw_of(nb);                                                                          \
          const int kb_ = kw_ < 3 ? kw_ : 0;                                                                  \
          const unsigned fw_ = (unsigned)(fl >> (((S) >> 1) * 32));                                           \
          const bool ok0_ = kw_ < 3 && ((fw_ >> ((((S) & 1) * 2 + 0) * 8 + kb_)) & 1u);                       \
          const bool ok1_ = kw_ < 3 && ((fw_ >> ((((S) & 1) * 2 + 1) * 8 + kb_)) & 1u);                       \
          const s16x4_t b0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + (ok0_ ? x_rd[S][0][nb] : x_zr[nb]))); \
          const s16x4_t b1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4_t)(bufp + (ok1_ ? x_rd[S][1][nb] : x_zr[nb]))); \
          fb[SET][nb] = __builtin_shufflevector(b0_, b1_, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        }                                                                                                     \
      }
#define YV4_FC_MFMA(SET)                                                                                      \
      {                                                                                                       \
        _Pragma("unroll") for (int nb = 0; nb < G::NBK; ++nb) {                                               \
          if (BF16)                                                                                           \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, fa[SET]),          \
                                                              __builtin_bit_cast(bf16x8_w, fb[SET][nb]), acc[nb], 0, 0, 0); \
          else                                                                                                \
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, fa[SET]),            \
                                                             __builtin_bit_cast(f16x8_w, fb[SET][nb]), acc[nb], 0, 0, 0);   \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
      }
      YV4_FC_LOAD(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_LOAD(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(0);
      YV4_FC_LOAD(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(1);
      YV4_FC_LOAD(1, 3);
      __builtin_amdgcn_sched_barrier(0);
      YV4_FC_MFMA(0);
      YV4_FC_MFMA(1);
#undef YV4_FC_MFMA
#undef YV4_FC_LOAD
      fl = fln;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the tail's out-of-range DMAs must land before the LDS goes

  if (!computes) return;
  // D[row = co][col]: row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31 -> (kw, ci) of the block
  const int ncol = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < G::NBK; ++nb) {
    int kw, ci;
    if (CIN == 16) { kw = 2 * nb + (ncol >> 4); ci = ncol & 15; }
    else if (CIN == 32) { kw = nb; ci = ncol; }
    else { kw = nb >> 1; ci = 32 * (nb & 1) + ncol; }
    if (kw >= 3 || ci >= p.Cin) continue;
    const int kcol = (kh * 3 + kw) * p.Cin + ci;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h5;
      if (co < p.Cout) {
        if (p.ws) p.ws[(size_t)chunk * p.ws_stride + (size_t)co * p.K + kcol] = acc[nb][e];
        else atomicAdd(&p.dw[(size_t)co * p.K + kcol], acc[nb][e]);
      }
    }
  }
}

// domain of conv_wgrad_fc_h16_kernel, and the channels per pixel it loads (0: not applicable)
static int wgrad_fc_cin(const yv4_conv_desc* d, int dtype) {
  static const int mode = YV4_ENV_INT("YV4_WGRAD_FC", 1);
  if (!mode || dtype == YV4_F32 || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W)
    return 0;
  if (d->Cout != 32 && d->Cout != 64) return 0;
  const long long M = (long long)d->N * d->H * d->W;
  if (M >= (1LL << 30) || M < 256LL * kFcRows * 8) return 0;           // at least eight slices for every CU
  if (d->Cin == 16 || d->Cin == 32 || d->Cin == 64) return d->Cin;
  // the stem: 8 weight channels against an image stored with 16 per pixel (the other 8 are read and dropped)
  if (d->Cin == 8 && d->x_coff + 16 <= d->x_cstride) return 16;
  return 0;
}

// The tiles of one reduction chunk read the same rows of dY and (shifted by a row) of the activation; workgroups go to the
// eight XCDs round-robin, so with the plain (tile, chunk) grid a chunk's tiles sit on different XCDs and every XCD's L2
// fetches those rows for itself.  With the mapping of wgrad_tile_chunk they share one L2.
static const int g_w3_xcd = YV4_ENV_INT("YV4_W3_XCD", 0);   // measured: 112 -> 115 / 113 -> 122 us on 128->128 @76 / 256->256 @38 -- off
static bool w3_xcd_map(long long tiles, long long chunks) { return g_w3_xcd && tiles >= 2 && chunks >= 16; }

// domain of conv_wgrad3x3_h16_kernel
static bool wgrad3x3_applies(const yv4_conv_desc* d, int dtype) {
  static const int mode = YV4_ENV_INT("YV4_WGRAD3", 1);
  return mode && dtype != YV4_F32 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->H &&
         d->Wo == d->W && (d->Cin & 127) == 0 && (long long)d->N * d->H * d->W < (1LL << 30);
}

// dw[i] += sum over chunks of slab_c[i], in a FIXED order: the deterministic tail of the weight gradient.
// A workgroup owns 16 float4 columns; its 16 chunk lanes q each add the slabs c = q, q + 16, q + 32, ... in ascending
// order (independent loads, 4 in flight), the 16 lane sums are then added in lane order.  (One thread per column
// walking all chunks serially was latency-bound on the 1x1 layers: 361 chunks of a 64 KB dW took 100 us.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int chunks, long long stride,
                                                           long long n, float* __restrict__ dw) {
  __shared__ float4 part[16][16];
  const long long n4 = n >> 2;
  const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
  const long long col = (long long)blockIdx.x * 16 + cl;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < n4) {
    const float* base = ws + 4 * col;
    int c = q;
    for (; c + 48 < chunks; c += 64) {
      const float4 b0 = *reinterpret_cast<const float4*>(base + (size_t)c * stride);
      const float4 b1 = *reinterpret_cast<const float4*>(base + (size_t)(c + 16) * stride);
      const float4 b2 = *reinterpret_cast<const float4*>(base + (size_t)(c + 32) * stride);
      const float4 b3 = *reinterpret_cast<const float4*>(base + (size_t)(c + 48) * stride);
      a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
      a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
      a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
    }
    for (; c < chunks; c += 16) {
      const float4 b = *reinterpret_cast<const float4*>(base + (size_t)c * stride);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  part[q][cl] = a;
  __syncthreads();
  if (q == 0 && col < n4) {
    float4 t = part[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 b = part[k][cl];
      t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    float4 d = reinterpret_cast<float4*>(dw)[col];
    d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
    reinterpret_cast<float4*>(dw)[col] = d;
  }
}

// ---------------------------------------------------------------------------------
// dst[n, 2y, 2x, c] = src[n, y, x, c], everything else 0  (dst is (N, 2H, 2W, C) dense NHWC).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H,
                                                      int W, int C4, int src_cs, int src_co) {
  const size_t total = (size_t)N * 2 * H * 2 * W * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % (2 * W));
    t /= 2 * W;
    const int y = (int)(t % (2 * H));
    const int n = (int)(t / (2 * H));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((x | y) & 1) == 0)
      v = *reinterpret_cast<const float4*>(src + ((size_t)(n * H + (y >> 1)) * W + (x >> 1)) * src_cs + src_co + c4 * 4);
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}

// ---------------------------------------------------------------------------------
// Train-mode BatchNorm.  x is an NHWC view (M rows, C channels).
//   stats:   per-channel sum and sum of squares, fp64 partials per workgroup -> atomics (double)
//   fwd:     z = (x - mean) * invstd * gamma + beta;  y = act(z) (+ residual)
//   bwd:     g = dy * act'(z);  dbeta = sum g;  dgamma = sum g * xhat;
//            dx = gamma * invstd * (g - dbeta/M - xhat * dgamma/M)
// act in {none, Mish, LeakyReLU, Swish}; Mish' as mmdet/ops/mish_cuda/src/mish.h:21-29.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH: {
      // mish.h:21-29 with sp = log1p(e^z), a = 1 + e^z, w = a^2 + 1:  tanh(sp) = (a^2 - 1) / (a^2 + 1) = 1 - 2 / w  and
      // (1 - tanh^2(sp)) * (1 - exp(-sp)) = (4 a^2 / w^2) * (e / a), so
      //     mish'(z) = 1 - 2 / w + 4 z a e / w^2
      // -- ONE reciprocal and one exp2 (hardware, 1 ulp each; quarter-rate instructions): |error| < 1e-6 against the libm
      // form, well inside the 1e-4 gradient budget.  The two BN-backward kernels are bound by exactly this arithmetic
      // (~35 issue slots per element at 16 lanes per SIMD and clock = their 0.6 ms on the 757 M-element layer); the
      // earlier form spent two reciprocals and ~6 more slots here.
      const float e = __builtin_amdgcn_exp2f(fminf(z, 20.f) * 1.44269504088896340736f);
      const float a = e + 1.f;
      const float iw = __builtin_amdgcn_rcpf(__builtin_fmaf(a, a, 1.f));
      const float g = __builtin_fmaf(4.f * (z * (a * e)), iw * iw, __builtin_fmaf(-2.f, iw, 1.f));
      return z >= 20.f ? 1.f : g;
    }
    case YV4_ACT_LEAKY: return z >= 0.f ? 1.f : slope;
    case YV4_ACT_SWISH: {
      const float s = 1.f / (1.f + expf(-z));
      return s + z * s * (1.f - s);
    }
    default: return 1.f;
  }
}
// Two channels at a time for the Mish passes of the BatchNorm kernels: the compiler does not pair the per-channel fp32
// arithmetic by itself (no v_pk_* in the scalar loops), and these kernels are bound by their VALU issue slots (a wave
// instruction takes four cycles on a 16-lane SIMD: ~30 slots per element = 0.6 ms on the 757 M-element layer, which is
// also its HBM time).  Every operation below is the scalar path's, done on a pair -- v_pk_mul / v_pk_add / v_pk_fma --
// so the results are bit for bit the scalar ones; the transcendentals stay one per element.
__device__ __forceinline__ f32x2_t splat2(float v) { f32x2_t r; r.x = v; r.y = v; return r; }
__device__ __forceinline__ f32x2_t mish_grad2(f32x2_t z) {
  f32x2_t zc;
  zc.x = fminf(z.x, 20.f); zc.y = fminf(z.y, 20.f);
  const f32x2_t t = zc * 1.44269504088896340736f;
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2_t a = e + 1.f;
  const f32x2_t w = __builtin_elementwise_fma(a, a, splat2(1.f));
  f32x2_t iw;
  iw.x = __builtin_amdgcn_rcpf(w.x); iw.y = __builtin_amdgcn_rcpf(w.y);
  f32x2_t g = __builtin_elementwise_fma(4.f * (z * (a * e)), iw * iw, __builtin_elementwise_fma(splat2(-2.f), iw, splat2(1.f)));
  g.x = z.x >= 20.f ? 1.f : g.x;
  g.y = z.y >= 20.f ? 1.f : g.y;
  return g;
}
// mish_fast_f32 on a pair, expression for expression: e = exp2(x log2 e), n = e (e + 2), (x n) / (n + 2), x itself from 20 on
__device__ __forceinline__ f32x2_t mish_fwd2(f32x2_t x) {
  const f32x2_t t = x * 1.44269504088896340736f;
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2_t n = e * (e + 2.f);
  const f32x2_t d = n + 2.f;
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  f32x2_t y = (x * n) * r;
  y.x = x.x >= 20.f ? x.x : y.x;
  y.y = x.y >= 20.f ? x.y : y.y;
  return y;
}
// (the forward of the fused BN + activation uses apply_act -- hardware exp2 / rcp Mish, < 2e-6 absolute from the
// libm form: with the libm form the kernel was VALU-bound, ~45 instructions per element at 2 bytes in, 2 out)
__device__ __forceinline__ float act_fwd_exact(float z, int act, float slope) {
  switch (act) {
    case YV4_ACT_MISH: return mish_f32(z);
    case YV4_ACT_LEAKY: return z >= 0.f ? z : z * slope;
    case YV4_ACT_SWISH: return z * sigmoid_f32(z);
    default: return z;
  }
}

#ifndef YV4_BN_RED_WAVES
#define YV4_BN_RED_WAVES 1
#endif
#ifndef YV4_BN_APPLY_WAVES
#define YV4_BN_APPLY_WAVES 1
#endif
constexpr int kBnRows = 8192;  // rows per workgroup at most (512 measured 1.2-1.5x slower on the >= 1 M-row maps:
                               // the per-workgroup LDS / global atomics then outweigh 32 KB of streaming)

// Thread map of the per-channel reductions: a row of the NHWC view is C4 = C/4 float4s; the
// workgroup's 256 threads cover rows_per_pass = 256 / C4 rows at a time (all threads busy and
// perfectly coalesced for every C4 <= 256; wider rows are walked in passes of 256 float4s).
struct RedMap {
  int cq0, cq_step, rsub, rstep;
  bool active;
};
__device__ __forceinline__ RedMap red_map(int C4) {
  RedMap m;
  if (C4 <= 256) {
    const int rpp = 256 / C4;
    m.active = (int)threadIdx.x < rpp * C4;
    m.cq0 = threadIdx.x % C4;
    m.cq_step = C4;          // one quad per thread
    m.rsub = threadIdx.x / C4;
    m.rstep = rpp;
  } else {
    m.active = true;
    m.cq0 = threadIdx.x;
    m.cq_step = 256;
    m.rsub = 0;
    m.rstep = 1;
  }
  return m;
}

// Block-level combine of per-thread partials (a: first C values, b: second C values) and one
// double atomic per channel per workgroup.  part[] lives in LDS: [2][C] doubles.
// det (yv4_set_deterministic): part[] is [2][2*C] 64-bit words -- hi words of (a | b), then their lo words (fx_add)
template <int SHIFT>
__device__ __forceinline__ void red_flush(double* part, int C, int c, const double (&a)[4], const double (&b)[4],
                                          bool active, int det) {
  if (active) {
    u64_t* w = reinterpret_cast<u64_t*>(part);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (det) {
        fx_add<SHIFT>(w + c + k, w + 2 * C + c + k, a[k]);
        fx_add<SHIFT>(w + C + c + k, w + 3 * C + c + k, b[k]);
      } else {
        atomicAdd(&part[c + k], a[k]);
        atomicAdd(&part[C + c + k], b[k]);
      }
    }
  }
}
// a workgroup's fixed-point words -> the global accumulator's (the sticky non-finite bit travels as an OR)
__device__ __forceinline__ void fx_merge(u64_t* ghi, u64_t* glo, u64_t h, u64_t l) {
  if (h) atomicAdd(ghi, h);
  if (l >> 63) atomicOr(glo, 1ull << 63);
  l &= ~(1ull << 63);
  if (l) atomicAdd(glo, l);
}
constexpr int kBnFloatRun = 16;  // unrolled iterations (x4 rows) a thread sums in fp32 before folding into its doubles

#ifndef YV4_BN_UNROLL
#define YV4_BN_UNROLL 4
#endif
constexpr int kBnUnroll = YV4_BN_UNROLL;    // independent row loads in flight per thread (the loops are latency-bound otherwise)
#ifndef YV4_BN_RED_UNROLL
#define YV4_BN_RED_UNROLL 2
#endif
constexpr int kBnRedUnroll = YV4_BN_RED_UNROLL;   // (4 and 8 measured 0.8 % / 3 % slower on the whole step: registers -> occupancy)

// rows per workgroup: enough workgroups to fill the chip (>= ~1024) but at most kBnRows rows each
static const int g_bn_rows_cap = YV4_ENV_INT("YV4_BN_ROWS", kBnRows);
static const int g_bn_min_wg = YV4_ENV_INT("YV4_BN_MINWG", 1024);
static inline int bn_rows_per_block(int64_t M) {
  int64_t r = (M + g_bn_min_wg - 1) / g_bn_min_wg;
  if (r < 32) r = 32;
  if (r > g_bn_rows_cap) r = g_bn_rows_cap;
  return (int)r;
}

// sums[c] += sum x, sums[C + c] += sum x^2   (double)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int64_t M, int C, int cs, int co,
                                                       double* __restrict__ sums, int rows_per_block, int det) {
  extern __shared__ double part[];   // [2][C]; det: [4][C] words
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < (det ? 4 : 2) * C; i += 256) part[i] = 0.0;
  __syncthreads();
  const RedMap mp = red_map(C4);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  if (mp.active) {
    for (int cq = mp.cq0; cq < C4; cq += mp.cq_step) {
      float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
      double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
      int it = 0;
      const T* col = x + co + cq * 4;
      for (int64_t rr = r0 + mp.rsub; rr < r1; rr += (int64_t)mp.rstep * kBnRedUnroll) {
        if (++it == kBnFloatRun) {       // bound the length of an fp32 running sum (64 rows)
          it = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) { ds[k] += fs[k]; dq[k] += fq[k]; fs[k] = 0.f; fq[k] = 0.f; }
        }
        float4 v[kBnRedUnroll];
#pragma unroll
        for (int u = 0; u < kBnRedUnroll; ++u) {
          const int64_t row = rr + (int64_t)u * mp.rstep;
          v[u] = row < r1 ? El<T>::ld4(col + row * cs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBnRedUnroll; ++u) {
          fs[0] += v[u].x; fs[1] += v[u].y; fs[2] += v[u].z; fs[3] += v[u].w;
          fq[0] += v[u].x * v[u].x; fq[1] += v[u].y * v[u].y; fq[2] += v[u].z * v[u].z; fq[3] += v[u].w * v[u].w;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { ds[k] += fs[k]; dq[k] += fq[k]; }
      red_flush<kFxStat>(part, C, cq * 4, ds, dq, true, det);
    }
  }
  __syncthreads();
  if (det) {      // sums: [hi words (2*C) | lo words (2*C)]
    const u64_t* w = reinterpret_cast<const u64_t*>(part);
    u64_t* g = reinterpret_cast<u64_t*>(sums);
    for (int i = threadIdx.x; i < 2 * C; i += 256) fx_merge(g + i, g + 2 * C + i, w[i], w[2 * C + i]);
    return;
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[i], part[i]);
}
// det: the words of [hi (n) | lo (n)] -> n doubles in place (consumers outside the library: SyncBN's all-reduce)
template <int SHIFT>
__global__ void fx_decode_kernel(double* __restrict__ buf, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64_t* w = reinterpret_cast<const u64_t*>(buf);
  buf[i] = fx_value<SHIFT>(w[i], w[n + i]);
}

// mean / biased var / invstd from the sums; running stats update (unbiased var, momentum)
// `rows`: optional device-resident row count (SyncBN: the all-reduced count travels with the sums)
// clear_work: the replicas are zeroed as they are read (a persistent statistics buffer is clean again for the next
// forward); zero_after: 4*C doubles cleared for the backward reduction of the same layer -- both replace memsets.
__global__ void bn_finalize_kernel(double* __restrict__ sums, int64_t M_host, int C, float eps, float momentum,
                                   float* mean, float* invstd, float* running_mean, float* running_var,
                                   const double* __restrict__ rows, int replicas, int clear_work,
                                   double* __restrict__ zero_after, int det) {
  // 256 threads = 32 channels x 8 replica lanes: a lane adds every 8th replica (independent loads in flight), the 8
  // lanes of a channel combine by shuffle.  (One thread per channel walking 64 replicas was a chain of 128 dependent
  // loads: 18 us per call, 2 ms of the bf16 train step over its 108 BatchNorms.)
  const int c = blockIdx.x * 32 + (threadIdx.x >> 3);
  const int rl = threadIdx.x & 7;
  double s1 = 0.0, s2 = 0.0;
  // Every load of a lane is issued before the first is used (the replica count is at most YV4_STATS_REPLICAS = 64: eight
  // per lane): as a loop over a run-time count the loads went out one iteration at a time behind the zeroing stores of the
  // iteration before -- a chain of eight memory round trips, 7.5 us per call and 0.85 ms of the bf16 train step.
  if (det) {
    // replica PAIRS of fixed-point words (stat_rep / bn_stats_kernel): integer sums over the pairs, any order
    u64_t h1 = 0, l1 = 0, h2 = 0, l2 = 0;
    if (c < C) {
      u64_t* w = reinterpret_cast<u64_t*>(sums);
      u64_t v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rl + 8 * k;
        const bool in = r < replicas / 2;
        u64_t* hp = w + (size_t)(2 * (in ? r : 0)) * 2 * C;
        u64_t* lp = hp + 2 * C;
        v[k][0] = in ? hp[c] : 0; v[k][1] = in ? lp[c] : 0; v[k][2] = in ? hp[C + c] : 0; v[k][3] = in ? lp[C + c] : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fx_fold(h1, l1, v[k][0], v[k][1]);
        fx_fold(h2, l2, v[k][2], v[k][3]);
      }
      if (clear_work) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rl + 8 * k;
          if (r < replicas / 2) {
            u64_t* hp = w + (size_t)(2 * r) * 2 * C;
            u64_t* lp = hp + 2 * C;
            hp[c] = 0; hp[C + c] = 0; lp[c] = 0; lp[C + c] = 0;
          }
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      fx_fold(h1, l1, __shfl_xor(h1, o), __shfl_xor(l1, o));
      fx_fold(h2, l2, __shfl_xor(h2, o), __shfl_xor(l2, o));
    }
    s1 = fx_value<kFxStat>(h1, l1);
    s2 = fx_value<kFxStat>(h2, l2);
  } else {
    if (c < C) {
      double a1[8], a2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = rl + 8 * k;
        const bool in = r < replicas;
        a1[k] = in ? sums[(size_t)r * 2 * C + c] : 0.0;
        a2[k] = in ? sums[(size_t)r * 2 * C + C + c] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1 += a1[k]; s2 += a2[k]; }      // (replica order rl, rl + 8, ...: as before)
      if (clear_work) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = rl + 8 * k;
          if (r < replicas) {
            sums[(size_t)r * 2 * C + c] = 0.0;
            sums[(size_t)r * 2 * C + C + c] = 0.0;
          }
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
  }
  if (c < C && zero_after && rl == 0) {      // 4*C words: [dbeta | dgamma] and, in deterministic mode, their lo words
#pragma unroll
    for (int k = 0; k < 4; ++k) zero_after[k * C + c] = 0.0;
  }
  if (c >= C || rl != 0) return;
  const double M = rows ? *rows : (double)M_host;
  const double m = s1 / M;
  double var = s2 / M - m * m;
  if (var < 0) var = 0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1 ? var * M / (M - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// totals of a conv epilogue's replicas as 2*C doubles (SyncBN: they are all-reduced before the finalize)
__global__ void stats_fold_kernel(double* __restrict__ sums, int C, int replicas, int clear_work, double* __restrict__ out,
                                  int det) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * C) return;
  if (det) {
    u64_t* w = reinterpret_cast<u64_t*>(sums);
    u64_t h = 0, l = 0;
    for (int r = 0; r < replicas / 2; ++r) {
      u64_t* hp = w + (size_t)(2 * r) * 2 * C + i;
      fx_fold(h, l, hp[0], hp[2 * C]);
      if (clear_work) { hp[0] = 0; hp[2 * C] = 0; }
    }
    out[i] = fx_value<kFxStat>(h, l);
  } else {
    double a = 0.0;
    for (int r = 0; r < replicas; ++r) {
      a += sums[(size_t)r * 2 * C + i];
      if (clear_work) sums[(size_t)r * 2 * C + i] = 0.0;
    }
    out[i] = a;
  }
}

struct BnArgs {
  const void* x; int x_cs, x_co;
  const float* mean; const float* invstd; const float* gamma; const float* beta;
  const void* res; int r_cs, r_co;
  void* y; int y_cs, y_co;
  const void* dy; int dy_cs, dy_co;
  void* dx; int dx_cs, dx_co;
  double* sums;      // bwd: [dbeta (C) | dgamma (C)]
  float* dgamma; float* dbeta;   // written by workgroup 0 of the apply pass
  int64_t M; int C; int act; float slope;
  int rows_per_block;
  int eval_mode;     // backward of an eval-mode BN (running statistics are constants): no mean/variance terms
  int64_t M_total;   // rows behind the statistics (= M, or the sum over ranks for SyncBN)
  const double* rows; // optional device-resident M_total
  int publish;       // the apply pass writes dgamma / dbeta from `sums` (not when `sums` were all-reduced)
  int red_cg;        // bn_act_bwd_reduce_kernel: channels per workgroup (grid.y groups)
  int det;           // `sums` holds fixed-point words: [hi (2*C) | lo (2*C)] (yv4_set_deterministic)
};

// entry i of the backward sums [dbeta (C) | dgamma (C)]
__device__ __forceinline__ double bn_sum(const BnArgs& p, int i) {
  if (!p.det) return p.sums[i];
  const u64_t* w = reinterpret_cast<const u64_t*>(p.sums);
  return fx_value<kFxGrad>(w[i], w[2 * p.C + i]);
}

// Elementwise passes use the reductions' thread map too: a thread keeps ONE channel group of V channels (its
// mean / invstd / gamma / beta live in registers) and walks rows -- no per-element index division,
// kBnUnroll independent row loads in flight.  V = 4 channels per thread (V = 8 for 16-bit rows: YV4_BN_VEC8=1).
// YV4_BN_NT (build-time, tools/ab_bn_nt.sh): 1 = the BatchNorm passes' row loads non-temporal, 2 = their stores.  Measured at
// YOLOv4-L 608 batch 64 bf16 on one box (profiles/r05_bn_nt_ab.txt): non-temporal STORES take the forward pass from 4.04 to
// 3.78 ms per step and the backward apply pass from 6.21 to 6.11, the train step from 1 194 to 1 199-1 204 images/s;
// non-temporal loads cost 4 % on both.  Default: stores only.
#ifndef YV4_BN_NT
#define YV4_BN_NT 2
#endif

template <typename T, int V> struct RowVec {
  typedef T raw __attribute__((ext_vector_type(V)));
  static __device__ __forceinline__ raw ld(const T* p) {
    if (YV4_BN_NT & 1) return __builtin_nontemporal_load(reinterpret_cast<const raw*>(p));
    return *reinterpret_cast<const raw*>(p);
  }
  static __device__ __forceinline__ void st(T* p, const float (&v)[V]) {
    raw o;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (T)v[k];
    if (YV4_BN_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<raw*>(p));
    else *reinterpret_cast<raw*>(p) = o;
  }
  static __device__ __forceinline__ raw zero() {
    raw o;
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (T)0.f;
    return o;
  }
};

template <typename T, int V>
__global__ __launch_bounds__(256, YV4_BN_APPLY_WAVES) void bn_act_fwd_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  const T* px = reinterpret_cast<const T*>(p.x);
  const T* pres = reinterpret_cast<const T*>(p.res);
  T* py = reinterpret_cast<T*>(p.y);
  const int CV = p.C / V;
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  if (!mp.active) return;
  for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
    const int c = cq * V;
    float mu[V], sa[V], be[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {       // z = (x - mu) * sa + be
      mu[k] = p.mean[c + k]; sa[k] = p.invstd[c + k] * p.gamma[c + k]; be[k] = p.beta[c + k];
    }
    for (int64_t rr = r0 + mp.rsub; rr < r1; rr += (int64_t)mp.rstep * kBnUnroll) {
      typename RV::raw v[kBnUnroll], rs[kBnUnroll];
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        const bool ok = row < r1;
        v[u] = ok ? RV::ld(px + row * p.x_cs + p.x_co + c) : RV::zero();
        rs[u] = (ok && pres) ? RV::ld(pres + row * p.r_cs + p.r_co + c) : RV::zero();
      }
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        if (row >= r1) continue;
        float o[V];
        if (p.act == YV4_ACT_MISH) {      // (uniform) pairs of channels: see mish_grad2
#pragma unroll
          for (int k = 0; k < V; k += 2) {
            f32x2_t x2, m2, s2, b2, r2;
            x2.x = (float)v[u][k]; x2.y = (float)v[u][k + 1];
            m2.x = mu[k]; m2.y = mu[k + 1]; s2.x = sa[k]; s2.y = sa[k + 1]; b2.x = be[k]; b2.y = be[k + 1];
            r2.x = (float)rs[u][k]; r2.y = (float)rs[u][k + 1];
            const f32x2_t y2 = mish_fwd2((x2 - m2) * s2 + b2) + r2;
            o[k] = y2.x; o[k + 1] = y2.y;
          }
        } else {
#pragma unroll
          for (int k = 0; k < V; ++k) o[k] = apply_act(((float)v[u][k] - mu[k]) * sa[k] + be[k], p.act, p.slope) + (float)rs[u][k];
        }
        RV::st(py + row * p.y_cs + p.y_co + c, o);
      }
    }
  }
}

// Grid: (row blocks, channel groups of p.red_cg channels).  Every workgroup ends with one double atomic per channel it
// covers; with ~1000 row blocks over ALL channels a small map (38 x 38 x 256 at batch 64: 47 MB) spent 12-14 us of its
// 43 us queueing ~1000 adds on each of its 512 addresses.  Splitting the channels over grid.y keeps the workgroup count
// (and the bytes in flight) and divides the adds per address by the number of groups; a group is >= 64 channels, so a
// workgroup still reads whole 128-byte lines of every row.
template <typename T, int V>
__global__ __launch_bounds__(256, YV4_BN_RED_WAVES) void bn_act_bwd_reduce_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  extern __shared__ double part[];   // [2][Cl]: dbeta | dgamma of this workgroup's channels
  const int cb = (int)blockIdx.y * p.red_cg;
  const int Cl = min(p.red_cg, p.C - cb);
  const T* px = reinterpret_cast<const T*>(p.x) + p.x_co + cb;
  const T* pdy = reinterpret_cast<const T*>(p.dy) + p.dy_co + cb;
  const int CV = Cl / V;
  for (int i = threadIdx.x; i < (p.det ? 4 : 2) * Cl; i += 256) part[i] = 0.0;
  __syncthreads();
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  if (mp.active) {
    for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
      const int c = cq * V;
      float mu[V], is[V], ga[V], be[V];
      // fp32 running sums over this thread's rows (at most rows_per_block / rows-per-pass, a few hundred terms):
      // double registers here cost a wave of occupancy (135 -> 119 VGPRs) and 35 % of the kernel's speed
      float db[V], dg[V];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        mu[k] = p.mean[cb + c + k]; is[k] = p.invstd[cb + c + k]; ga[k] = p.gamma[cb + c + k]; be[k] = p.beta[cb + c + k];
        db[k] = 0.f; dg[k] = 0.f;
      }
      for (int64_t rr = r0 + mp.rsub; rr < r1; rr += (int64_t)mp.rstep * kBnRedUnroll) {
        typename RV::raw xv[kBnRedUnroll], gv[kBnRedUnroll];
#pragma unroll
        for (int u = 0; u < kBnRedUnroll; ++u) {
          const int64_t row = rr + (int64_t)u * mp.rstep;
          const bool ok = row < r1;
          xv[u] = ok ? RV::ld(px + row * p.x_cs + c) : RV::zero();
          gv[u] = ok ? RV::ld(pdy + row * p.dy_cs + c) : RV::zero();   // zero beyond r1 -> contributes nothing
        }
        if (p.act == YV4_ACT_MISH) {      // (uniform) pairs of channels: see mish_grad2
#pragma unroll
          for (int u = 0; u < kBnRedUnroll; ++u) {
#pragma unroll
            for (int k = 0; k < V; k += 2) {
              f32x2_t x2, g2, m2, i2, a2, b2, db2, dg2;
              x2.x = (float)xv[u][k]; x2.y = (float)xv[u][k + 1];
              g2.x = (float)gv[u][k]; g2.y = (float)gv[u][k + 1];
              m2.x = mu[k]; m2.y = mu[k + 1]; i2.x = is[k]; i2.y = is[k + 1];
              a2.x = ga[k]; a2.y = ga[k + 1]; b2.x = be[k]; b2.y = be[k + 1];
              db2.x = db[k]; db2.y = db[k + 1]; dg2.x = dg[k]; dg2.y = dg[k + 1];
              const f32x2_t xh2 = (x2 - m2) * i2;
              const f32x2_t gg = g2 * mish_grad2(xh2 * a2 + b2);
              db2 = db2 + gg;
              dg2 = dg2 + gg * xh2;
              db[k] = db2.x; db[k + 1] = db2.y; dg[k] = dg2.x; dg[k + 1] = dg2.y;
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < kBnRedUnroll; ++u) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
              const float xhat = ((float)xv[u][k] - mu[k]) * is[k];
              const float g = (float)gv[u][k] * act_grad(xhat * ga[k] + be[k], p.act, p.slope);
              db[k] += g;
              dg[k] += g * xhat;
            }
          }
        }
      }
#pragma unroll
      for (int h = 0; h < V; h += 4) {
        const double ddb[4] = {db[h], db[h + 1], db[h + 2], db[h + 3]}, ddg[4] = {dg[h], dg[h + 1], dg[h + 2], dg[h + 3]};
        red_flush<kFxGrad>(part, Cl, c + h, ddb, ddg, true, p.det);
      }
    }
  }
  __syncthreads();
  if (p.det) {
    const u64_t* w = reinterpret_cast<const u64_t*>(part);
    u64_t* g = reinterpret_cast<u64_t*>(p.sums);
    for (int i = threadIdx.x; i < Cl; i += 256) {
      fx_merge(g + cb + i, g + 2 * p.C + cb + i, w[i], w[2 * Cl + i]);
      fx_merge(g + p.C + cb + i, g + 3 * p.C + cb + i, w[Cl + i], w[3 * Cl + i]);
    }
    return;
  }
  for (int i = threadIdx.x; i < Cl; i += 256) {
    atomicAdd(&p.sums[cb + i], part[i]);
    atomicAdd(&p.sums[p.C + cb + i], part[Cl + i]);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256, YV4_BN_APPLY_WAVES) void bn_act_bwd_apply_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  const T* px = reinterpret_cast<const T*>(p.x);
  const T* pdy = reinterpret_cast<const T*>(p.dy);
  T* pdx = reinterpret_cast<T*>(p.dx);
  const int CV = p.C / V;
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  if (blockIdx.x == 0 && p.publish) {   // the reduction kernel has completed (stream order): publish dbeta / dgamma as fp32
    for (int i = threadIdx.x; i < p.C; i += 256) {
      const double sb = bn_sum(p, i), sg = bn_sum(p, p.C + i);
      if (p.publish == 2) {             // accumulate into existing gradients (the parameter's .grad itself)
        p.dbeta[i] += (float)sb;
        p.dgamma[i] += (float)sg;
      } else {
        p.dbeta[i] = (float)sb;
        p.dgamma[i] = (float)sg;
      }
    }
  }
  if (!mp.active) return;
  const double invM = 1.0 / (p.rows ? *p.rows : (double)p.M_total);
  for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
    const int c = cq * V;
    // dx = k1 * (g - dbm - xhat * dgm),  xhat = (x - mu) * is,  z = xhat * ga + be
    float mu[V], is[V], ga[V], be[V], k1[V], dbm[V], dgm[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      mu[k] = p.mean[c + k]; is[k] = p.invstd[c + k]; ga[k] = p.gamma[c + k]; be[k] = p.beta[c + k];
      k1[k] = ga[k] * is[k];
      dbm[k] = p.eval_mode ? 0.f : (float)(bn_sum(p, c + k) * invM);
      dgm[k] = p.eval_mode ? 0.f : (float)(bn_sum(p, p.C + c + k) * invM);
    }
    for (int64_t rr = r0 + mp.rsub; rr < r1; rr += (int64_t)mp.rstep * kBnUnroll) {
      typename RV::raw xv[kBnUnroll], gv[kBnUnroll];
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        const bool ok = row < r1;
        xv[u] = ok ? RV::ld(px + row * p.x_cs + p.x_co + c) : RV::zero();
        gv[u] = ok ? RV::ld(pdy + row * p.dy_cs + p.dy_co + c) : RV::zero();
      }
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        if (row >= r1) continue;
        float o[V];
        // (scalar on purpose: the paired form of the other two passes costs this one 15 registers and, bound by its
        // 6 bytes per element as it is, 4 % of its speed -- tools/bn_bench.py --kernels, same box)
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float xhat = ((float)xv[u][k] - mu[k]) * is[k];
          const float g = (float)gv[u][k] * act_grad(xhat * ga[k] + be[k], p.act, p.slope);
          o[k] = p.eval_mode ? k1[k] * g : k1[k] * (g - dbm[k] - xhat * dgm[k]);
        }
        RV::st(pdx + row * p.dx_cs + p.dx_co + c, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// The three BatchNorm + Mish row passes for 16-bit maps, second form (round 5).  The general kernels above were measured
// at 5.3 / 5.5 / 7.1 ms per bf16 step and priced at "~35 issue slots per element"; the disassembly says otherwise: the
// Mish derivative is 15 packed fp32 operations, 4 transcendentals (8 issue cycles each on this part, not 16), 6 scalar
// compare / select / min and 4 conversions per PAIR of elements = ~150 issue cycles per pair and wave, 2.7 ms per pass on
// the whole chip -- and 4 bytes per element at 5.5 TB/s are 4.2 ms.  The passes run at neither roof but at most of their
// SUM: a wave loads its rows, waits, computes, stores, and 4-5 waves per SIMD do not cover one another's waits.  Here:
//   * the row loop is software-pipelined: the loads of rows i + U .. i + 2U are in flight while rows i .. i + U are
//     computed (two register sets, the loop unrolled by two so that no set is ever copied);
//   * per-channel constants are folded (z = A x + B with A = gamma * invstd, B = beta - mean * A; the backward's
//     dx = k1 g + (c1 x + c0)): 2-5 registers per channel instead of 3-7, one fma instead of subtract + multiply + fma;
//   * Mish and its derivative clamp the exponent's argument instead of selecting the asymptote afterwards (for z >= 20 the
//     expressions round to z and to 1 by themselves): two v_min per pair instead of two compares and two selects;
//   * no run-time activation switch inside the loops (Mish only; anything else stays on the general kernels).
// 16-bit outputs are the fp32 expression rounded once; against the general kernels they differ by the re-association of
// the affine map (<= 1 ulp of the 16-bit type, tests/test_gpu_train_ops.py::test_bn16_*).  fp32 maps never come here.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ f32x2_t exp_clamped2(f32x2_t z) {       // e^min(z, 20)
  f32x2_t zc;
  zc.x = fminf(z.x, 20.f); zc.y = fminf(z.y, 20.f);
  const f32x2_t t = zc * 1.44269504088896340736f;
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  return e;
}
__device__ __forceinline__ f32x2_t mish_fwd2c(f32x2_t z) {          // z n / (n + 2), n = e (e + 2)
  const f32x2_t e = exp_clamped2(z);
  const f32x2_t n = e * (e + 2.f);
  const f32x2_t d = n + 2.f;
  f32x2_t r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return z * (n * r);
}
__device__ __forceinline__ f32x2_t mish_grad2c(f32x2_t z) {         // 1 - u + z a e u^2, a = 1 + e, u = 2 / (a^2 + 1)
  f32x2_t zc;
  zc.x = fminf(z.x, 20.f); zc.y = fminf(z.y, 20.f);
  const f32x2_t t = zc * 1.44269504088896340736f;
  f32x2_t e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2_t a = e + 1.f;
  const f32x2_t w = __builtin_elementwise_fma(a, a, splat2(1.f));
  f32x2_t iw;
  iw.x = __builtin_amdgcn_rcpf(w.x); iw.y = __builtin_amdgcn_rcpf(w.y);
  const f32x2_t u = iw + iw;
  return __builtin_elementwise_fma(zc * (a * e), u * u, splat2(1.f) - u);
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void bn16_fwd_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  typedef typename RV::raw raw;
  const T* px = reinterpret_cast<const T*>(p.x);
  const T* pres = reinterpret_cast<const T*>(p.res);
  T* py = reinterpret_cast<T*>(p.y);
  const int CV = p.C / V;
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  if (!mp.active) return;
  const bool has_res = pres != nullptr;
  const int64_t step = (int64_t)mp.rstep * U;
  for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
    const int c = cq * V;
    f32x2_t A[V / 2], B[V / 2];
#pragma unroll
    for (int k = 0; k < V; k += 2) {
      A[k / 2].x = p.invstd[c + k] * p.gamma[c + k];
      A[k / 2].y = p.invstd[c + k + 1] * p.gamma[c + k + 1];
      B[k / 2].x = p.beta[c + k] - p.mean[c + k] * A[k / 2].x;
      B[k / 2].y = p.beta[c + k + 1] - p.mean[c + k + 1] * A[k / 2].y;
    }
    // (no range checks in here: a select between a loaded value and zero makes the wave wait for the load where it is
    // ISSUED, which is exactly what the pipeline is there to avoid -- the main loop only runs on whole stages)
    auto load = [&](int64_t rr, raw (&xv)[U], raw (&rv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        xv[u] = RV::ld(px + row * p.x_cs + p.x_co + c);
        rv[u] = has_res ? RV::ld(pres + row * p.r_cs + p.r_co + c) : RV::zero();
      }
    };
    auto work = [&](int64_t rr, const raw (&xv)[U], const raw (&rv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        float o[V];
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          f32x2_t x2, r2;
          x2.x = (float)xv[u][k]; x2.y = (float)xv[u][k + 1];
          r2.x = (float)rv[u][k]; r2.y = (float)rv[u][k + 1];
          const f32x2_t y2 = mish_fwd2c(__builtin_elementwise_fma(x2, A[k / 2], B[k / 2])) + r2;
          o[k] = y2.x; o[k + 1] = y2.y;
        }
        RV::st(py + row * p.y_cs + p.y_co + c, o);
      }
    };
    raw xa[U], ra[U], xb[U], rb[U];
    int64_t rr = r0 + mp.rsub;
    const int64_t span = (int64_t)(2 * U - 1) * mp.rstep;      // a double stage starting at rr touches rows rr .. rr + span
    if (rr + span < r1) {
      load(rr, xa, ra);
      for (;;) {
        load(rr + step, xb, rb);
        work(rr, xa, ra);
        const int64_t nx = rr + 2 * step;
        const bool more = nx + span < r1;
        load(more ? nx : rr, xa, ra);      // (always issued -- past the end it re-reads this stage: a branch here makes the
                                             // compiler wait for EVERY load at the join, the next stage's included)
        work(rr + step, xb, rb);
        rr = nx;
        if (!more) break;
      }
    }
    for (; rr < r1; rr += mp.rstep) {                           // the rows that do not fill a double stage
      const raw xv = RV::ld(px + rr * p.x_cs + p.x_co + c);
      const raw rv = has_res ? RV::ld(pres + rr * p.r_cs + p.r_co + c) : RV::zero();
      float o[V];
#pragma unroll
      for (int k = 0; k < V; k += 2) {
        f32x2_t x2, r2;
        x2.x = (float)xv[k]; x2.y = (float)xv[k + 1];
        r2.x = (float)rv[k]; r2.y = (float)rv[k + 1];
        const f32x2_t y2 = mish_fwd2c(__builtin_elementwise_fma(x2, A[k / 2], B[k / 2])) + r2;
        o[k] = y2.x; o[k + 1] = y2.y;
      }
      RV::st(py + rr * p.y_cs + p.y_co + c, o);
    }
  }
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void bn16_bwd_reduce_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  typedef typename RV::raw raw;
  extern __shared__ double part[];   // [2][Cl]: dbeta | dgamma of this workgroup's channels (det: [4][Cl] words)
  const int cb = (int)blockIdx.y * p.red_cg;
  const int Cl = min(p.red_cg, p.C - cb);
  const T* px = reinterpret_cast<const T*>(p.x) + p.x_co + cb;
  const T* pdy = reinterpret_cast<const T*>(p.dy) + p.dy_co + cb;
  const int CV = Cl / V;
  for (int i = threadIdx.x; i < (p.det ? 4 : 2) * Cl; i += 256) part[i] = 0.0;
  __syncthreads();
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  const int64_t step = (int64_t)mp.rstep * U;
  if (mp.active) {
    for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
      const int c = cq * V;
      // z = A x + B, xhat = I x + J
      f32x2_t A[V / 2], B[V / 2], I[V / 2], J[V / 2], db[V / 2], dg[V / 2];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float is = p.invstd[cb + c + k], mu = p.mean[cb + c + k], ga = p.gamma[cb + c + k], be = p.beta[cb + c + k];
        const float a_ = is * ga;
        if (k & 1) { A[k / 2].y = a_; B[k / 2].y = be - mu * a_; I[k / 2].y = is; J[k / 2].y = -mu * is; }
        else { A[k / 2].x = a_; B[k / 2].x = be - mu * a_; I[k / 2].x = is; J[k / 2].x = -mu * is; }
      }
#pragma unroll
      for (int k = 0; k < V / 2; ++k) { db[k] = splat2(0.f); dg[k] = splat2(0.f); }
      auto load = [&](int64_t rr, raw (&xv)[U], raw (&gv)[U]) {       // (whole stages only: see bn16_fwd_kernel)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t row = rr + (int64_t)u * mp.rstep;
          xv[u] = RV::ld(px + row * p.x_cs + c);
          gv[u] = RV::ld(pdy + row * p.dy_cs + c);
        }
      };
      auto row_terms = [&](const raw& xv, const raw& gv) {
#pragma unroll
        for (int k = 0; k < V; k += 2) {
          f32x2_t x2, g2;
          x2.x = (float)xv[k]; x2.y = (float)xv[k + 1];
          g2.x = (float)gv[k]; g2.y = (float)gv[k + 1];
          const f32x2_t gg = g2 * mish_grad2c(__builtin_elementwise_fma(x2, A[k / 2], B[k / 2]));
          db[k / 2] = db[k / 2] + gg;
          dg[k / 2] = __builtin_elementwise_fma(gg, __builtin_elementwise_fma(x2, I[k / 2], J[k / 2]), dg[k / 2]);
        }
      };
      auto work = [&](const raw (&xv)[U], const raw (&gv)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) row_terms(xv[u], gv[u]);
      };
      raw xa[U], ga[U], xb[U], gb[U];
      int64_t rr = r0 + mp.rsub;
      const int64_t span = (int64_t)(2 * U - 1) * mp.rstep;
      if (rr + span < r1) {
        load(rr, xa, ga);
        for (;;) {
          load(rr + step, xb, gb);
          work(xa, ga);
          const int64_t nx = rr + 2 * step;
          const bool more = nx + span < r1;
          load(more ? nx : rr, xa, ga);      // (always issued -- past the end it re-reads this stage: a branch here makes the
                                             // compiler wait for EVERY load at the join, the next stage's included)
          work(xb, gb);
          rr = nx;
          if (!more) break;
        }
      }
      for (; rr < r1; rr += mp.rstep) row_terms(RV::ld(px + rr * p.x_cs + c), RV::ld(pdy + rr * p.dy_cs + c));
#pragma unroll
      for (int h = 0; h < V; h += 4) {
        const double ddb[4] = {db[h / 2].x, db[h / 2].y, db[h / 2 + 1].x, db[h / 2 + 1].y};
        const double ddg[4] = {dg[h / 2].x, dg[h / 2].y, dg[h / 2 + 1].x, dg[h / 2 + 1].y};
        red_flush<kFxGrad>(part, Cl, c + h, ddb, ddg, true, p.det);
      }
    }
  }
  __syncthreads();
  if (p.det) {
    const u64_t* w = reinterpret_cast<const u64_t*>(part);
    u64_t* g = reinterpret_cast<u64_t*>(p.sums);
    for (int i = threadIdx.x; i < Cl; i += 256) {
      fx_merge(g + cb + i, g + 2 * p.C + cb + i, w[i], w[2 * Cl + i]);
      fx_merge(g + p.C + cb + i, g + 3 * p.C + cb + i, w[Cl + i], w[3 * Cl + i]);
    }
    return;
  }
  for (int i = threadIdx.x; i < Cl; i += 256) {
    atomicAdd(&p.sums[cb + i], part[i]);
    atomicAdd(&p.sums[p.C + cb + i], part[Cl + i]);
  }
}

template <typename T, int V, int U>
__global__ __launch_bounds__(256) void bn16_bwd_apply_kernel(BnArgs p) {
  typedef RowVec<T, V> RV;
  typedef typename RV::raw raw;
  const T* px = reinterpret_cast<const T*>(p.x);
  const T* pdy = reinterpret_cast<const T*>(p.dy);
  T* pdx = reinterpret_cast<T*>(p.dx);
  const int CV = p.C / V;
  const RedMap mp = red_map(CV);
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_block;
  const int64_t r1 = r0 + p.rows_per_block < p.M ? r0 + p.rows_per_block : p.M;
  if (blockIdx.x == 0 && p.publish) {   // the reduction kernel has completed (stream order): publish dbeta / dgamma as fp32
    for (int i = threadIdx.x; i < p.C; i += 256) {
      const double sb = bn_sum(p, i), sg = bn_sum(p, p.C + i);
      if (p.publish == 2) {
        p.dbeta[i] += (float)sb;
        p.dgamma[i] += (float)sg;
      } else {
        p.dbeta[i] = (float)sb;
        p.dgamma[i] = (float)sg;
      }
    }
  }
  if (!mp.active) return;
  const double invM = 1.0 / (p.rows ? *p.rows : (double)p.M_total);
  const int64_t step = (int64_t)mp.rstep * U;
  for (int cq = mp.cq0; cq < CV; cq += mp.cq_step) {
    const int c = cq * V;
    // dx = k1 (g - dbm - xhat dgm) = k1 g + (c1 x + c0),  c1 = -k1 dgm invstd,  c0 = -k1 dbm + k1 dgm mean invstd
    f32x2_t A[V / 2], B[V / 2], K1[V / 2], C0[V / 2], C1[V / 2];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float is = p.invstd[c + k], mu = p.mean[c + k], ga = p.gamma[c + k], be = p.beta[c + k];
      const float a_ = is * ga;
      const float dbm = p.eval_mode ? 0.f : (float)(bn_sum(p, c + k) * invM);
      const float dgm = p.eval_mode ? 0.f : (float)(bn_sum(p, p.C + c + k) * invM);
      const float k1 = a_, c1 = -(k1 * dgm) * is, c0 = -(k1 * dbm) - c1 * mu;
      if (k & 1) { A[k / 2].y = a_; B[k / 2].y = be - mu * a_; K1[k / 2].y = k1; C0[k / 2].y = c0; C1[k / 2].y = c1; }
      else { A[k / 2].x = a_; B[k / 2].x = be - mu * a_; K1[k / 2].x = k1; C0[k / 2].x = c0; C1[k / 2].x = c1; }
    }
    auto load = [&](int64_t rr, raw (&xv)[U], raw (&gv)[U]) {         // (whole stages only: see bn16_fwd_kernel)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = rr + (int64_t)u * mp.rstep;
        xv[u] = RV::ld(px + row * p.x_cs + p.x_co + c);
        gv[u] = RV::ld(pdy + row * p.dy_cs + p.dy_co + c);
      }
    };
    auto one_row = [&](int64_t row, const raw& xv, const raw& gv) {
      float o[V];
#pragma unroll
      for (int k = 0; k < V; k += 2) {
        f32x2_t x2, g2;
        x2.x = (float)xv[k]; x2.y = (float)xv[k + 1];
        g2.x = (float)gv[k]; g2.y = (float)gv[k + 1];
        const f32x2_t gg = g2 * mish_grad2c(__builtin_elementwise_fma(x2, A[k / 2], B[k / 2]));
        const f32x2_t d2 = __builtin_elementwise_fma(K1[k / 2], gg, __builtin_elementwise_fma(C1[k / 2], x2, C0[k / 2]));
        o[k] = d2.x; o[k + 1] = d2.y;
      }
      RV::st(pdx + row * p.dx_cs + p.dx_co + c, o);
    };
    auto work = [&](int64_t rr, const raw (&xv)[U], const raw (&gv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) one_row(rr + (int64_t)u * mp.rstep, xv[u], gv[u]);
    };
    raw xa[U], ga[U], xb[U], gb[U];
    int64_t rr = r0 + mp.rsub;
    const int64_t span = (int64_t)(2 * U - 1) * mp.rstep;
    if (rr + span < r1) {
      load(rr, xa, ga);
      for (;;) {
        load(rr + step, xb, gb);
        work(rr, xa, ga);
        const int64_t nx = rr + 2 * step;
        const bool more = nx + span < r1;
        load(more ? nx : rr, xa, ga);      // (always issued -- past the end it re-reads this stage: a branch here makes the
                                             // compiler wait for EVERY load at the join, the next stage's included)
        work(rr + step, xb, gb);
        rr = nx;
        if (!more) break;
      }
    }
    for (; rr < r1; rr += mp.rstep)
      one_row(rr, RV::ld(px + rr * p.x_cs + p.x_co + c), RV::ld(pdy + rr * p.dy_cs + p.dy_co + c));
  }
}

// ---------------------------------------------------------------------------------
// SPP backward (darknetcsp.py:176-181,203-206,222-226: cat([x, mp5(x), mp9(x), mp13(x)])):
//   dx[p] = dcat[0][p] + sum over k in {5,9,13}, over output positions q whose window argmax is p,
//   of dcat[k][q].
// One thread owns (n, y, x, 4 channels) as an OUTPUT position: it rescans the 13x13 window of the
// saved input once in row-major order, tracking the first maximum of the nested 5 / 9 / 13 windows
// (torch's max_pool2d keeps the first maximum in scan order), and scatters its three gradients with
// float atomics into the fp32 accumulator dx (N, H, W, C dense, zero on entry), plus its own
// identity-branch gradient.  Replaces three ATen max_pool2d backward passes + three adds.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void spp_pool_bwd_kernel(const T* __restrict__ xcat, int x_cs, int x_co,
                                                           const T* __restrict__ dcat, int d_cs, int d_co,
                                                           float* __restrict__ dx, int N, int H, int W, int C) {
  const int C4 = C >> 2;
  const size_t total = (size_t)N * H * W * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float ninf = -__builtin_huge_valf();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const T* base = xcat + (size_t)n * H * W * x_cs + x_co + c4 * 4;
    float m[3][4];
    int am[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) { m[k][u] = ninf; am[k][u] = y * W + x; }
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const int ady = dy < 0 ? -dy : dy;
      for (int dxx = -6; dxx <= 6; ++dxx) {
        const int xx = x + dxx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const int adx = dxx < 0 ? -dxx : dxx;
        const int rad = ady > adx ? ady : adx;
        const float4 v4 = El<T>::ld4(base + ((size_t)yy * W + xx) * x_cs);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const int pos = yy * W + xx;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (v[u] > m[2][u]) { m[2][u] = v[u]; am[2][u] = pos; }
          if (rad <= 4 && v[u] > m[1][u]) { m[1][u] = v[u]; am[1][u] = pos; }
          if (rad <= 2 && v[u] > m[0][u]) { m[0][u] = v[u]; am[0][u] = pos; }
        }
      }
    }
    const T* g = dcat + ((size_t)(n * H + y) * W + x) * d_cs + d_co + c4 * 4;
    float* dxn = dx + (size_t)n * H * W * C + c4 * 4;
    const float4 g0 = El<T>::ld4(g);
    const float gi[4] = {g0.x, g0.y, g0.z, g0.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) atomicAdd(dxn + (size_t)(y * W + x) * C + u, gi[u]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 gk = El<T>::ld4(g + (k + 1) * C);
      const float gv[4] = {gk.x, gk.y, gk.z, gk.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) atomicAdd(dxn + (size_t)am[k][u] * C + u, gv[u]);
    }
  }
}

// The same scatter for the maps an SPP block actually sees (19x19 at 608 px): everything in LDS, and the window
// argmax found by CASCADED 5x5 pools instead of a 13x13 scan per pixel.
//   * Every element becomes a KEY: (order-preserving bits of the value) : (all-ones - position).  The maximum key of a
//     window is its largest value and, among equal values, the smallest position -- the first hit of the row-major scan
//     `v > best` that torch's pooling (and the kernel above) performs.  Keys make the argmax a plain associative,
//     idempotent max, so pool9 = pool5 o pool5 and pool13 = pool5 o pool9 exactly (windows clipped at the border), and
//     each 5x5 pool separates into a row pass and a column pass: 30 LDS reads per element for the three pools instead
//     of 169 global loads and 507 compare/select pairs (the round-2 form: 0.99 ms at batch 64 x 512 channels).
//   * one workgroup = one image x CG channels (8 for 16-bit keys, 4 for 64-bit keys of fp32 values): three key planes (in, row-pass, out -- rotated through the cascade)
//     and the fp32 accumulator plane, H*W x CG each; the three pool gradients go to the accumulator by LDS atomics,
//     the identity branch by a plain add, and dx is written once.
template <typename T> struct SppKey;
template <> struct SppKey<float> {
  typedef unsigned long long K;
  static constexpr int CG = 4;
  static __device__ __forceinline__ K make(float v, int pos) {
    unsigned b = __float_as_uint(v);
    if (b == 0x80000000u) b = 0u;                                     // -0 == +0 for `>`
    b ^= (b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
    return ((K)b << 32) | (K)(0xFFFFFFFFu - (unsigned)pos);
  }
  static __device__ __forceinline__ int pos(K k) { return (int)(0xFFFFFFFFu - (unsigned)k); }
};
template <typename T> struct SppKey {                                 // _Float16 / __bf16
  typedef unsigned K;
  static constexpr int CG = 8;
  static __device__ __forceinline__ K make(T v, int pos) {
    unsigned b = (unsigned)__builtin_bit_cast(unsigned short, v);
    if (b == 0x8000u) b = 0u;
    b ^= (b & 0x8000u) ? 0xFFFFu : 0x8000u;
    return (b << 16) | (0xFFFFu - (unsigned)pos);
  }
  static __device__ __forceinline__ int pos(K k) { return (int)(0xFFFFu - (k & 0xFFFFu)); }
};

constexpr int kSppItems = 16;      // (position, channel) items per thread: H*W*CG <= 4096 (the 64 KB LDS bound of the launch)
template <typename T>
__global__ __launch_bounds__(256) void spp_pool_bwd_lds_kernel(const T* __restrict__ xcat, int x_cs, int x_co,
                                                               const T* __restrict__ dcat, int d_cs, int d_co,
                                                               float* __restrict__ dx, int H, int W, int C, int det) {
  // det (yv4_set_deterministic): the accumulator plane holds 64-bit FIXED-POINT integers with one exponent for the
  // workgroup -- 2^40 / (the power of two above the largest |gradient| it will add, found by an integer max) -- so the
  // scatter's atomics are integer adds and the plane's value does not depend on their order.  A non-finite gradient
  // anywhere in the block makes the block's outputs NaN (the step is skipped by the loss scaler either way).
  typedef SppKey<T> SK;
  typedef typename SK::K K;
  constexpr int CG = SK::CG;
  extern __shared__ __attribute__((aligned(16))) unsigned char spp_raw[];
  const int HW = H * W;
  K* ka = reinterpret_cast<K*>(spp_raw);             // [HW][CG]
  K* kb = ka + (size_t)HW * CG;
  K* kc = kb + (size_t)HW * CG;
  float* acc = reinterpret_cast<float*>(kc + (size_t)HW * CG);
  long long* acc64 = reinterpret_cast<long long*>(acc);
  __shared__ unsigned smax;
  if (det && threadIdx.x == 0) smax = 0u;
  if (det) __syncthreads();
  unsigned gmax = 0u;
  const int n = blockIdx.y;
  const int cg0 = blockIdx.x * CG;
  const int nc = min(CG, C - cg0);
  const int items = HW * CG;
  const T* xb = xcat + (size_t)n * HW * x_cs + x_co + cg0;
  const T* gb = dcat + (size_t)n * HW * d_cs + d_co + cg0;
  const FastDiv fd_w = make_fastdiv((unsigned)W);
  // a thread keeps the same items (i = tid + 256 j) through every pass: their coordinates and their three pool gradients
  // are fetched once, all loads in flight together
  float g[3][kSppItems];
  float gid0[kSppItems];           // (deterministic mode only)
  short iy[kSppItems], ix[kSppItems];
#pragma unroll
  for (int j = 0; j < kSppItems; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int pos = i / CG, c = i - pos * CG;
    const int y = fd_div(pos, fd_w);
    iy[j] = (short)y;
    ix[j] = (short)(pos - y * W);
    const bool ok = i < items && c < nc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g[k][j] = ok ? (float)gb[(size_t)pos * d_cs + (size_t)(k + 1) * C + c] : 0.f;
      gmax = max(gmax, __float_as_uint(g[k][j]) & 0x7fffffffu);
    }
    if (i < items) {
      ka[i] = ok ? SK::make(xb[(size_t)pos * x_cs + c], pos) : (K)0;
      const float gid = ok ? (float)gb[(size_t)pos * d_cs + c] : 0.f;     // the identity branch's gradient
      if (det) { gid0[j] = gid; gmax = max(gmax, __float_as_uint(gid) & 0x7fffffffu); }
      else acc[i] = gid;
    }
  }
  double fx_scale = 1.0;
  bool fx_bad = false;
  if (det) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = max(gmax, (unsigned)__shfl_xor((int)gmax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&smax, gmax);
    __syncthreads();
    const unsigned mb = smax;
    fx_bad = mb >= 0x7f800000u;
    fx_scale = __builtin_ldexp(1.0, 166 - (int)(mb >> 23));      // |g| < 2^(e - 126)  ->  |g * scale| < 2^40
#pragma unroll
    for (int j = 0; j < kSppItems; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < items) acc64[i] = fx_bad ? 0ll : (long long)__builtin_rint((double)gid0[j] * fx_scale);
    }
  }
  __syncthreads();
  K* src = ka; K* tmp = kb; K* out = kc;
#pragma unroll 1
  for (int k = 0; k < 3; ++k) {
    // row pass: tmp(y, x) = max src(y, x-2 .. x+2)
#pragma unroll
    for (int j = 0; j < kSppItems; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < items) {
        const int c = i & (CG - 1), y = iy[j], x = ix[j];
        const K* row = src + (size_t)y * W * CG + c;
        K m = row[x * CG];
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
          if (d == 0) continue;
          const int xx = min(max(x + d, 0), W - 1);              // a clamped neighbour repeats an element of the window
          const K v = row[xx * CG];
          m = v > m ? v : m;
        }
        tmp[i] = m;
      }
    }
    __syncthreads();
    // column pass + scatter of this pool's gradient to its argmax
#pragma unroll
    for (int j = 0; j < kSppItems; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < items) {
        const int c = i & (CG - 1), y = iy[j], x = ix[j];
        const K* col = tmp + (size_t)x * CG + c;
        K m = col[(size_t)y * W * CG];
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
          if (d == 0) continue;
          const int yy = min(max(y + d, 0), H - 1);
          const K v = col[(size_t)yy * W * CG];
          m = v > m ? v : m;
        }
        out[i] = m;
        const float gv = k == 0 ? g[0][j] : (k == 1 ? g[1][j] : g[2][j]);
        if (c < nc) {
          if (det) {
            if (!fx_bad) atomicAdd(reinterpret_cast<u64_t*>(&acc64[SK::pos(m) * CG + c]), (u64_t)(long long)__builtin_rint((double)gv * fx_scale));
          } else {
            atomicAdd(&acc[SK::pos(m) * CG + c], gv);
          }
        }
      }
    }
    __syncthreads();
    K* t = src; src = out; out = t;                  // the pooled keys feed the next 5x5 pool
  }
  float* o = dx + (size_t)n * HW * C + cg0;
#pragma unroll
  for (int j = 0; j < kSppItems; ++j) {
    const int i = threadIdx.x + 256 * j;
    if (i < items) {
      const int pos = i / CG, c = i - pos * CG;
      if (c < nc) o[(size_t)pos * C + c] = !det ? acc[i] : (fx_bad ? __builtin_nanf("") : (float)((double)acc64[i] / fx_scale));
    }
  }
}

// Conv weight -> the kernels' packed operand in ONE pass (cast included): rows x (KHo*KWo*ICp) with K ordered
// (kh, kw, channel), zero-padded channels; output tap (kh, kw) reads source tap (kh0 + kh*kh_step, kw0 + kw*kw_step).
// transpose = 0: rows = Cout, channel = Cin (forward operand); 1: rows = Cin, channel = Cout -- with the taps
// mirrored (kh0 = KH-1, step -1) the operand of the data gradient (ATen needs flip + transpose + contiguous + cast = 3
// launches per conv per step for it), with a tap subset the operand of one parity class of a stride-2 data gradient
// (list-indexing the taps cost two host-to-device index uploads and two gather kernels per class).  The source is addressed through its element strides,
// so contiguous and channels_last parameters both go without a copy.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, long long s_co, long long s_ci,
                                                          long long s_kh, long long s_kw, int Cout, int Cin, int KHo, int KWo,
                                                          int kh0, int kh_step, int kw0, int kw_step, int tf, int ICp,
                                                          T* __restrict__ dst, int nrows) {
  // one output row (r, kh, kw) of ICp channels per workgroup iteration: two small divides per row, none per element
  const int IC = tf ? Cout : Cin;
  const int taps = KHo * KWo;
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int r = row / taps;
    const int tap = row - r * taps;
    const int kh = tap / KWo, kw = tap - kh * KWo;
    const float* src = w + (kh0 + kh * kh_step) * s_kh + (kw0 + kw * kw_step) * s_kw + (tf ? r * s_ci : r * s_co);
    const long long s_ic = tf ? s_co : s_ci;
    T* d = dst + (size_t)row * ICp;
    for (int ic = threadIdx.x; ic < ICp; ic += 256) d[ic] = (T)(ic < IC ? src[ic * s_ic] : 0.f);
  }
}

// The same pass over a TABLE of weights in one launch (yv4_pack_weights_multi): workgroup b serves the descriptor whose
// [first_block, first_block + nblocks) range holds b, rows_per_block output rows of it.
//
// An output row (r, kh, kw) runs over the channel ic; in the SOURCE (an fp32 (Cout, Cin, KH, KW) parameter, normally
// contiguous) the element sits at r*s_r + ic*s_ic + tap offset, and whichever of the forward operand (s_ic = KH*KW) and
// the data-gradient operand (s_ic = Cin*KH*KW) is packed, neighbouring ic are 36 bytes or kilobytes apart: reading row
// by row (round 2) moved 4 bytes per 64- or 128-byte line touched and took 0.87 ms per YOLOv4-L step (64 M parameters,
// both operands).  Here a workgroup stages a box of the source -- NR rows r x ICc channels x every tap the descriptor
// uses -- in LDS, walking the source in ITS order (taps fastest, then whichever of r / ic has the smaller stride), and
// writes the output rows from LDS with the channel across lanes.
constexpr int kPackStage = 9216;       // floats staged per pass (36 KB)

template <typename T>
__device__ __forceinline__ void pack_rows(const yv4_pack_desc& d, int row0, int row1, float* stage) {
  const int tf = d.transpose;
  const int IC = tf ? d.Cout : d.Cin;
  const int ICp = (IC + d.pad_to - 1) / d.pad_to * d.pad_to;
  const int taps = d.KHo * d.KWo;
  const long long s_ic = tf ? d.s_co : d.s_ci, s_r = tf ? d.s_ci : d.s_co;
  T* dst = reinterpret_cast<T*>(d.dst);
  // bounding box of the source taps the descriptor reads
  const int khl = d.kh0 + (d.KHo - 1) * d.kh_step, kwl = d.kw0 + (d.KWo - 1) * d.kw_step;
  const int khmin = min(d.kh0, khl), kwmin = min(d.kw0, kwl);
  const int nbh = abs(khl - d.kh0) + 1, nbw = abs(kwl - d.kw0) + 1;
  const int TB = nbh * nbw, TBs = TB | 1;                    // odd LDS pitch per (r, ic): channel-strided reads hit all banks
  const int rA = row0 / taps, rB = (row1 - 1) / taps;        // rows r touched (inclusive)
  const bool ic_inner = s_ic <= s_r;
  int NR, ICc;
  if (ic_inner) {
    if (ICp * TBs <= kPackStage) { ICc = ICp; NR = min(rB - rA + 1, kPackStage / (ICp * TBs)); }
    else { NR = 1; ICc = (kPackStage / TBs) & ~7; }
  } else {
    NR = min(rB - rA + 1, max(8, 64 / TB));                  // >= 256 contiguous source bytes per ic
    ICc = min(ICp, (kPackStage / (NR * TBs)) & ~7);
  }
  constexpr int VEC = sizeof(T) == 2 ? 2 : 1;                // 16-bit outputs are stored in pairs
  const FastDiv fd_tb = make_fastdiv((unsigned)TB), fd_bw = make_fastdiv((unsigned)nbw), fd_kwo = make_fastdiv((unsigned)d.KWo),
                fd_taps = make_fastdiv((unsigned)taps);
  const int tid = threadIdx.x;
  for (int r0 = rA; r0 <= rB; r0 += NR) {
    const int nr = min(NR, rB - r0 + 1);
    const FastDiv fd_nr = make_fastdiv((unsigned)nr);
    const int orow0 = max(row0, r0 * taps), orow1 = min(row1, (r0 + nr) * taps);
    for (int c0 = 0; c0 < ICp; c0 += ICc) {
      const int cn = min(ICc, ICp - c0);
      const FastDiv fd_cn = make_fastdiv((unsigned)cn);
      __syncthreads();                                       // the previous pass has been written out
      const int total = nr * cn * TB;
      // eight independent loads in flight per thread (the staging is latency-bound otherwise)
      for (int e0 = tid; e0 < total; e0 += 256 * 8) {
        float v[8];
        int li[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + 256 * u;
          v[u] = 0.f;
          li[u] = -1;
          if (e < total) {
            const int q = fd_div(e, fd_tb), t = e - q * TB;
            int rl, cl;
            if (ic_inner) { rl = fd_div(q, fd_cn); cl = q - rl * cn; }
            else { cl = fd_div(q, fd_nr); rl = q - cl * nr; }
            const int bh = fd_div(t, fd_bw), bw = t - bh * nbw;
            const int ic = c0 + cl;
            li[u] = (rl * cn + cl) * TBs + t;
            if (ic < IC) v[u] = d.w[(long long)(r0 + rl) * s_r + (long long)ic * s_ic + (khmin + bh) * d.s_kh + (kwmin + bw) * d.s_kw];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (li[u] >= 0) stage[li[u]] = v[u];
      }
      __syncthreads();
      const int cv = cn / VEC;
      const FastDiv fd_cv = make_fastdiv((unsigned)cv);
      const int wtotal = (orow1 - orow0) * cv;
      for (int i = tid; i < wtotal; i += 256) {
        const int ro = fd_div(i, fd_cv), pc = i - ro * cv;
        const int row = orow0 + ro;
        const int r = fd_div(row, fd_taps), tap = row - r * taps;
        const int kh = fd_div(tap, fd_kwo), kw = tap - kh * d.KWo;
        const int tb = (d.kh0 + kh * d.kh_step - khmin) * nbw + (d.kw0 + kw * d.kw_step - kwmin);
        const float* sp = stage + ((r - r0) * cn + pc * VEC) * TBs + tb;
        T* o = dst + (size_t)row * ICp + c0 + pc * VEC;
        if constexpr (VEC == 2) {
          union { T h[2]; unsigned u; } pk;
          pk.h[0] = (T)sp[0];
          pk.h[1] = (T)sp[TBs];
          *reinterpret_cast<unsigned*>(o) = pk.u;
        } else {
          o[0] = (T)sp[0];
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const yv4_pack_desc* __restrict__ table, int n) {
  __shared__ float stage[kPackStage];
  // binary search of the descriptor (uniform per workgroup)
  int lo = 0, hi = n - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const yv4_pack_desc d = table[lo];
  const int R = d.transpose ? d.Cin : d.Cout;
  const int nrows = R * d.KHo * d.KWo;
  const int row0 = (b - d.first_block) * d.rows_per_block;
  const int row1 = row0 + d.rows_per_block < nrows ? row0 + d.rows_per_block : nrows;
  if (row0 >= row1) return;
  switch (d.dtype) {
    case YV4_F32: pack_rows<float>(d, row0, row1, stage); break;
    case YV4_F16: pack_rows<_Float16>(d, row0, row1, stage); break;
    default: pack_rows<__bf16>(d, row0, row1, stage); break;
  }
}

__global__ void sums_to_float_kernel(const double* __restrict__ sums, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)sums[i];
}

static inline unsigned ew_grid_t(size_t work_items) {
  size_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

// Backward of the nearest resample by an INTEGER factor (yolo_neck_csp.py:213-219: F.interpolate(scale 2) into the concat
// buffer): dx[n, sy, sx, c] = the sum of the fy x fx gradient pixels that read it, fp32 sum, one rounding.  The gradient is a
// channel slice of the concat buffer's gradient (dy_cs / dy_co).
template <typename T>
__global__ __launch_bounds__(256) void resample_nearest_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int Hs,
                                                                   int Ws, int fy, int fx, int C4, int dy_cs, int dy_co) {
  const size_t total = (size_t)N * Hs * Ws * C4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int Wd = Ws * fx, Hd = Hs * fy;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    size_t t = i / C4;
    const int sx = (int)(t % Ws);
    t /= Ws;
    const int sy = (int)(t % Hs);
    const int n = (int)(t / Hs);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < fy; ++j)
      for (int k = 0; k < fx; ++k) {
        const float4 v = El<T>::ld4(dy + ((size_t)(n * Hd + sy * fy + j) * Wd + sx * fx + k) * dy_cs + dy_co + c4 * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    El<T>::st4(dx + i * 4, a);
  }
}

}  // namespace yv4

using namespace yv4;

// dtype-dispatching bodies shared by the fp32 entries and their _h16 forms ------------------------
#define YV4_DISPATCH_T(dtype, CALL)                    \
  switch (dtype) {                                     \
    case YV4_F32: { typedef float T; CALL; } break;    \
    case YV4_F16: { typedef _Float16 T; CALL; } break; \
    default: { typedef __bf16 T; CALL; } break;        \
  }

// ... and the vector width of the BN row passes: 4 channels for fp32, 8 for 16-bit operands whose strides allow it
#define YV4_DISPATCH_TV(dtype, v8, CALL)                                          \
  switch (dtype) {                                                                \
    case YV4_F32: { typedef float T; constexpr int V = 4; CALL; } break;          \
    case YV4_F16: { typedef _Float16 T; if (v8) { constexpr int V = 8; CALL; } else { constexpr int V = 4; CALL; } } break; \
    default: { typedef __bf16 T; if (v8) { constexpr int V = 8; CALL; } else { constexpr int V = 4; CALL; } } break;        \
  }
// (ablation switch, off by default: 8 channels per thread -- 16-byte accesses on 16-bit rows -- measured no faster on the
// forward pass and 20 % SLOWER on the backward apply pass over YOLOv4-L's shapes, tools/bn_bench.py --kernels: the
// passes are bound by bytes in flight per CU, which the extra registers reduce)
static const bool g_bn_vec8 = YV4_ENV_INT("YV4_BN_VEC8", 0) == 1;
// the pipelined 16-bit Mish passes (bn16_*): on / off, channels per thread (4 or 8) and rows per pipeline stage
static const int g_bn16 = YV4_ENV_INT("YV4_BN16", 1);
static const int g_bn16_v = YV4_ENV_INT("YV4_BN16_V", 4);
#ifndef YV4_BN16_U
#define YV4_BN16_U 2
#endif
#define YV4_DISPATCH_H16V(dtype, v8, CALL)                                                                      \
  if ((dtype) == YV4_F16) { typedef _Float16 T; if (v8) { constexpr int V = 8; CALL; } else { constexpr int V = 4; CALL; } } \
  else { typedef __bf16 T; if (v8) { constexpr int V = 8; CALL; } else { constexpr int V = 4; CALL; } }

// test / ablation switch: route 16-bit inputs through the widening fp32-MFMA kernel instead of the
// 16-bit MFMA one (YV4_WGRAD_WIDEN=1 in the environment)
static const bool g_wgrad_widen = YV4_ENV_INT("YV4_WGRAD_WIDEN", 0) == 1;
// measurement switch: the XCD-aware (tile, chunk) mapping of the 16-bit weight-gradient kernels (wgrad_tile_chunk)
static const bool g_wgrad_xcd = YV4_ENV_INT("YV4_WGRAD_XCD", 1) != 0;

// split of the M reduction into chunks (shared by the launch and by yv4_conv_wgrad_workspace)
static void wgrad_chunks(const yv4_conv_desc* d, int dtype, long long* chunks, long long* rows) {
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const int K = d->KH * d->KW * d->Cin;
  if (!g_wgrad_widen && wgrad_fc_cin(d, dtype)) {
    // one round of workgroups (two per CU; one for Cin 64), every one with at least four slices (wgrad_fc_cin)
    long long ch = wgrad_fc_cin(d, dtype) == 64 ? 256 : 512;
    long long rw = (M + ch - 1) / ch;
    rw = (rw + kFcRows - 1) / kFcRows * kFcRows;
    *rows = rw;
    *chunks = (M + rw - 1) / rw;
    return;
  }
  if (!g_wgrad_widen && wgrad3x3_applies(d, dtype)) {
    // one 8-wave workgroup per CU.  Measured (tools/wgrad_bench.py --det, batch 64): ONE full round of (co tile, kh, ci
    // tile, chunk) workgroups beats two (half the slab traffic and epilogues: 160 vs 179 us on 128->128 @76) unless the
    // tiles leave more than ~10 % of the CUs idle (512->1024 @19: 96 tiles x 2 chunks = 192 workgroups, 403 vs 342 us);
    // never one workgroup beyond a full round (it costs a whole round).
    const long long tl = (long long)((d->Cout + 127) / 128) * 3 * (d->Cin / 128);
    static const int cus = YV4_ENV_INT("YV4_WGRAD3_CUS", 256);
    long long ch = cus / tl;
    if (ch < 1 || tl * ch * 10 < (long long)cus * 9) ch = (2 * cus) / tl;
    const long long mx = (M + 8 * kW3Rows - 1) / (8 * kW3Rows);
    if (ch > mx) ch = mx;
    if (ch < 1) ch = 1;
    if (ch > 65535) ch = 65535;
    // XCD-aware mapping (wgrad_tile_chunk): a chunk's tiles on ONE XCD need the chunk count in whole groups of eight
    if (w3_xcd_map(tl, ch)) ch = ch / 8 * 8;
    long long rw = (M + ch - 1) / ch;
    rw = (rw + kW3Rows - 1) / kW3Rows * kW3Rows;
    *rows = rw;
    *chunks = (M + rw - 1) / rw;
    return;
  }
  if (dtype != YV4_F32 && !g_wgrad_widen) {
    const long long tl = (long long)((K + kWhTile - 1) / kWhTile) * ((d->Cout + kWhTile - 1) / kWhTile);
    // Chunks of the reduction: ONE round of two workgroups per CU and never a workgroup more (513 workgroups take two
    // rounds).  Measured over YOLOv4-L at batch 64 (tools/wgrad_bench.py --det, measure build, YV4_WGRAD_WGS): 512 beats
    // 1024 on every 1x1 layer (half the partial-sum slabs: 512->256 @38 56 -> 47 us) and on the stride-2 layers, network
    // 505 -> 525 TFLOP/s.  A dW of more than half a round of tiles cannot fill one round: then the chunk count with the
    // least (rounds / chunks), the smallest within 20 % of it (512->1024 s2 @38: 288 tiles x 3 chunks, 383 -> 312 us).
    static const int wg_target = YV4_ENV_INT("YV4_WGRAD_WGS", 512);
    static const int min_slices = YV4_ENV_INT("YV4_WGRAD_MINSL", 16);
    const long long mx = (M + min_slices * kWhRows - 1) / (min_slices * kWhRows);   // at least min_slices per chunk
    long long ch = wg_target / tl;
    if (2 * tl > wg_target) {
      const long long chmax = 4 * wg_target / tl > 1 ? 4 * wg_target / tl : 1;
      double best = 1e30;
      for (long long c = 1; c <= chmax; ++c) {
        const double sc = (double)((tl * c + wg_target - 1) / wg_target) / (double)c;
        if (sc < best) best = sc;
      }
      for (long long c = 1; c <= chmax; ++c)
        if ((double)((tl * c + wg_target - 1) / wg_target) / (double)c <= 1.2 * best) { ch = c; break; }
    }
    if (ch > mx) ch = mx;
    if (ch < 1) ch = 1;
    if (tl >= 2 && ch >= 16 && d->Cout >= 128) ch = ch / 8 * 8;   // whole groups of 8 chunks, one per XCD (wgrad_tile_chunk)
    if (ch > 65528) ch = 65528;
    long long rw = (M + ch - 1) / ch;
    rw = (rw + kWhRows - 1) / kWhRows * kWhRows;
    *rows = rw;
    *chunks = (M + rw - 1) / rw;
    return;
  }
  const long long tiles = (long long)((K + 63) / 64) * ((d->Cout + 63) / 64);
  long long ch = (256 * 4 + tiles - 1) / tiles;            // ~4 workgroups per CU, at least 8 slices each
  const long long mx = (M + 8 * kWgRows - 1) / (8 * kWgRows);
  if (ch > mx) ch = mx;
  if (ch < 1) ch = 1;
  if (ch > 65535) ch = 65535;
  long long rw = (M + ch - 1) / ch;
  rw = (rw + kWgRows - 1) / kWgRows * kWgRows;
  *rows = rw;
  *chunks = (M + rw - 1) / rw;
}

static int wgrad_impl(const yv4_conv_desc* d, int dtype, const void* x, const void* dy, float* dw, void* stream,
                      float* ws = nullptr, size_t ws_bytes = 0) {
  YV4_REQUIRE(d && x && dy && dw, "wgrad: null argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "wgrad: dtype must be f32, f16 or bf16");
  const int al = dtype == YV4_F32 ? 4 : 8;
  const int es = dtype == YV4_F32 ? 4 : 2;
  YV4_REQUIRE(d->Cin % al == 0 && d->x_cstride % al == 0 && d->x_coff % al == 0,
              "wgrad: input channels/stride/offset must be multiples of %d", al);
  YV4_REQUIRE(d->Cout % al == 0 && d->y_cstride % al == 0 && d->y_coff % al == 0,
              "wgrad: dY channels/stride/offset must be multiples of %d", al);
  YV4_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 64 && d->stride > 0, "wgrad: bad kernel/stride");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  YV4_REQUIRE(Ho == d->Ho && Wo == d->Wo, "wgrad: Ho/Wo do not match the geometry");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  const long long xb = (long long)d->N * d->H * d->W * d->x_cstride * es, db = M * d->y_cstride * es;
  YV4_REQUIRE(M < (1LL << 31) && xb < 0xFFFFFFF0LL && db < 0xFFFFFFF0LL, "wgrad: tensors of 4 GiB or more are not supported");
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.x_cs = d->x_cstride; a.x_co = d->x_coff; a.dy_cs = d->y_cstride; a.dy_co = d->y_coff;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin;
  a.fd_hw = make_fastdiv((unsigned)(d->Ho * d->Wo));
  a.fd_wo = make_fastdiv((unsigned)d->Wo);
  long long ch = 1, rw = M;
  wgrad_chunks(d, dtype, &ch, &rw);
  a.rows_per_chunk = (int)rw;
  const long long dw_elems = (long long)a.Cout * a.K;
  if (ws && ch > 1) {
    YV4_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)dw & 15) == 0 && dw_elems % 4 == 0,
                "wgrad: workspace / dw must be 16-byte aligned and Cout*K a multiple of 4");
    YV4_REQUIRE(ws_bytes >= (size_t)ch * dw_elems * sizeof(float), "wgrad: workspace too small (%zu bytes for %lld chunks)",
                ws_bytes, ch);
    a.ws = ws;
    a.ws_stride = dw_elems;
  }
  auto finish = [&]() -> int {
    if (!a.ws) return YV4_OK;
    const long long g = (dw_elems / 4 + 15) / 16;
    if (g > 0x7fffffffLL) { set_error("wgrad: dW too large"); return YV4_E_INVALID; }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a.ws, (int)ch,
                       a.ws_stride, dw_elems, dw);
    YV4_CHECK_LAUNCH("conv_wgrad reduce");
    return YV4_OK;
  };
  // (the second forms range-check 32-bit byte OFFSETS: both maps below 3 GB.  Beyond that the product takes the generic
  // 16-bit kernel further down; the first forms of the two special kernels exist in the measurement build only)
  static const int fcv2 = YV4_ENV_INT("YV4_WFC_V2", 1);
  const bool fc_v2_ok = fcv2 && xb < 0xC0000000LL && db < 0xC0000000LL;
#ifdef YV4_MEASURE
  const bool fc_any = true;
#else
  const bool fc_any = fc_v2_ok;
#endif
  if (const int fc = (g_wgrad_widen || !fc_any) ? 0 : wgrad_fc_cin(d, dtype)) {
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    static const int fc_ablate = YV4_ENV_INT("YV4_WFC_ABLATE", 0);
    a.ablate = fc_ablate;
    const bool use_v2 = fc_v2_ok;
#ifdef YV4_MEASURE
#define YV4_FC_FIRST(CI, CO)                                                                                         \
  else {                                                                                                             \
    static LdsAttrOnce once_b, once_h;                                                                               \
    constexpr size_t lds = FcGeom<CI, CO>::Lds;                                                                      \
    if (int rc = ensure_dyn_lds(once_b, reinterpret_cast<const void*>(conv_wgrad_fc_h16_kernel<true, CI, CO>), lds, "conv_wgrad_fc_h16")) return rc;  \
    if (int rc = ensure_dyn_lds(once_h, reinterpret_cast<const void*>(conv_wgrad_fc_h16_kernel<false, CI, CO>), lds, "conv_wgrad_fc_h16")) return rc; \
    if (dtype == YV4_BF16) hipLaunchKernelGGL((conv_wgrad_fc_h16_kernel<true, CI, CO>), dim3((unsigned)ch), dim3(kFcThreads), lds, hs, a, (unsigned)xb, (unsigned)db); \
    else hipLaunchKernelGGL((conv_wgrad_fc_h16_kernel<false, CI, CO>), dim3((unsigned)ch), dim3(kFcThreads), lds, hs, a, (unsigned)xb, (unsigned)db);                  \
  }
#else
#define YV4_FC_FIRST(CI, CO)
#endif
#define YV4_FC_LAUNCH(CI, CO)                                                                                        \
  { if (use_v2) {                                                                                                    \
    static LdsAttrOnce once_b2, once_h2;                                                                             \
    constexpr size_t lds2 = FcGeom<CI, CO>::Lds + FcGeom<CI, CO>::NBuf * 192;                                        \
    if (int rc = ensure_dyn_lds(once_b2, reinterpret_cast<const void*>(conv_wgrad_fc_v2_h16_kernel<true, CI, CO>), lds2, "conv_wgrad_fc_v2_h16")) return rc;  \
    if (int rc = ensure_dyn_lds(once_h2, reinterpret_cast<const void*>(conv_wgrad_fc_v2_h16_kernel<false, CI, CO>), lds2, "conv_wgrad_fc_v2_h16")) return rc; \
    if (dtype == YV4_BF16) hipLaunchKernelGGL((conv_wgrad_fc_v2_h16_kernel<true, CI, CO>), dim3((unsigned)ch), dim3(kFcThreads), lds2, hs, a, (unsigned)xb, (unsigned)db); \
    else hipLaunchKernelGGL((conv_wgrad_fc_v2_h16_kernel<false, CI, CO>), dim3((unsigned)ch), dim3(kFcThreads), lds2, hs, a, (unsigned)xb, (unsigned)db);                  \
  } YV4_FC_FIRST(CI, CO) }
    if (a.Cout == 32) {
      if (fc == 16) YV4_FC_LAUNCH(16, 32) else if (fc == 32) YV4_FC_LAUNCH(32, 32) else YV4_FC_LAUNCH(64, 32)
    } else {
      if (fc == 16) YV4_FC_LAUNCH(16, 64) else if (fc == 32) YV4_FC_LAUNCH(32, 64) else YV4_FC_LAUNCH(64, 64)
    }
#undef YV4_FC_LAUNCH
#undef YV4_FC_FIRST
    YV4_CHECK_LAUNCH("conv_wgrad_fc_h16");
    return finish();
  }
  static const int w3v2 = YV4_ENV_INT("YV4_W3V2", 1);
  const bool w3_v2_ok = w3v2 && xb < 0xC0000000LL && db < 0xC0000000LL;
#ifdef YV4_MEASURE
  const bool w3_any = true;
#else
  const bool w3_any = w3_v2_ok;
#endif
  if (!g_wgrad_widen && w3_any && wgrad3x3_applies(d, dtype)) {
    const long long tl = (long long)((a.Cout + 127) / 128) * 3 * (a.Cin / 128);
#ifdef YV4_MEASURE
    static LdsAttrOnce once3b, once3h;
    if (int rc = ensure_dyn_lds(once3b, reinterpret_cast<const void*>(conv_wgrad3x3_h16_kernel<true>), (size_t)kW3Lds, "conv_wgrad3x3_h16")) return rc;
    if (int rc = ensure_dyn_lds(once3h, reinterpret_cast<const void*>(conv_wgrad3x3_h16_kernel<false>), (size_t)kW3Lds, "conv_wgrad3x3_h16")) return rc;
#endif
    a.tiles = (int)tl;
    a.chunks = (int)ch;
    a.xcd_map = w3_xcd_map(tl, ch) && tl * (ch + 8) < (1LL << 31) ? 1 : 0;
    static const int w3_ablate = YV4_ENV_INT("YV4_W3_ABLATE", 0);
    a.ablate = w3_ablate;
    const dim3 grid3 = wgrad_grid(tl, ch, a.xcd_map);
    // (the second form range-checks 32-bit byte OFFSETS: both maps well below 4 GB, so that a row in front of the map --
    // a wrapped offset -- can never fall below a limit)
    if (w3_v2_ok) {
      static LdsAttrOnce once3vb, once3vh;
      if (int rc = ensure_dyn_lds(once3vb, reinterpret_cast<const void*>(conv_wgrad3x3_v2_h16_kernel<true>), (size_t)kW3LdsV2, "conv_wgrad3x3_v2_h16")) return rc;
      if (int rc = ensure_dyn_lds(once3vh, reinterpret_cast<const void*>(conv_wgrad3x3_v2_h16_kernel<false>), (size_t)kW3LdsV2, "conv_wgrad3x3_v2_h16")) return rc;
#ifdef YV4_MEASURE
      static const int w3abl = YV4_ENV_INT("YV4_W3V2_ABL", 0);
#define YV4_W3ABL(N)                                                                                                   \
      if (w3abl == N && dtype == YV4_BF16) {                                                                           \
        static LdsAttrOnce once_abl;                                                                                   \
        if (int rc = ensure_dyn_lds(once_abl, reinterpret_cast<const void*>(conv_wgrad3x3_v2_h16_kernel<true, N>), (size_t)kW3LdsV2, "w3v2 abl")) return rc; \
        hipLaunchKernelGGL((conv_wgrad3x3_v2_h16_kernel<true, N>), grid3, dim3(kW3Threads), (size_t)kW3LdsV2,         \
                           reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db);                      \
        YV4_CHECK_LAUNCH("w3v2 abl");                                                                                  \
        return finish();                                                                                               \
      }
      YV4_W3ABL(1) YV4_W3ABL(2) YV4_W3ABL(3) YV4_W3ABL(4) YV4_W3ABL(8) YV4_W3ABL(12) YV4_W3ABL(5) YV4_W3ABL(13) YV4_W3ABL(15) YV4_W3ABL(16)
#undef YV4_W3ABL
#endif
      if (dtype == YV4_BF16)
        hipLaunchKernelGGL(conv_wgrad3x3_v2_h16_kernel<true>, grid3, dim3(kW3Threads), (size_t)kW3LdsV2,
                           reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db);
      else
        hipLaunchKernelGGL(conv_wgrad3x3_v2_h16_kernel<false>, grid3, dim3(kW3Threads), (size_t)kW3LdsV2,
                           reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db);
      YV4_CHECK_LAUNCH("conv_wgrad3x3_v2_h16");
      return finish();
    }
#ifdef YV4_MEASURE
    if (dtype == YV4_BF16)
      hipLaunchKernelGGL(conv_wgrad3x3_h16_kernel<true>, grid3, dim3(kW3Threads), (size_t)kW3Lds,
                         reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db);
    else
      hipLaunchKernelGGL(conv_wgrad3x3_h16_kernel<false>, grid3, dim3(kW3Threads), (size_t)kW3Lds,
                         reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db);
    YV4_CHECK_LAUNCH("conv_wgrad3x3_h16");
    return finish();
#endif
  }
  if (dtype != YV4_F32 && !g_wgrad_widen) {
    // 16-bit MFMA form: 128 x 128 tiles of dW, 64-row slices
    a.tiles_k = (a.K + kWhTile - 1) / kWhTile;
    const int tc = (a.Cout + kWhTile - 1) / kWhTile;
    const long long tl = (long long)a.tiles_k * tc;
    // slice buffers: 2 x two workgroups per CU.  (3 to 5 buffers for ONE workgroup per CU -- more bytes in flight, no
    // queue drain -- are 20-60 % slower on every layer, network 505 -> 408 TFLOP/s: with one wave per SIMD nothing
    // covers a wave's transposed-read -> MFMA chain.  Those instantiations exist in the measure build only.)
    static const int nbuf = YV4_ENV_INT("YV4_WGRAD_NBUF", 2);
    const size_t ldsh = (size_t)nbuf * 2 * kWhRows * 256;
    a.tiles = (int)tl;
    a.chunks = (int)ch;
    // (not for the few-channel layers at 304 / 608 pixels -- Cout < 128, half-empty dY tiles, pure streaming: inside the
    // training step they ran 25 % slower with it, tools/train_timeline.py)
    a.xcd_map = g_wgrad_xcd && tl >= 2 && ch >= 16 && a.Cout >= 128 && tl * (ch + 8) < (1LL << 31) ? 1 : 0;
    const dim3 grid = wgrad_grid(tl, ch, a.xcd_map);
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    static const int whv2 = YV4_ENV_INT("YV4_WGRAD_V2", 1);
    const bool linear = d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0;
    // (second form: offsets are range-checked as 32-bit byte offsets, window origins travel as 16-bit coordinates.  Not for
    // the few-channel windowed layers -- 32 -> 64 s2 @608 streams 2.3 GB through 288 columns of dW and is bound by its
    // bytes: the table's LDS round trip in front of every slice's DMA cost it 7 %, 836 -> 894 us)
    static const int whv2_min_cin = YV4_ENV_INT("YV4_WGRAD_V2_MINCIN", 64);
    if (whv2 && nbuf == 2 && xb < 0xC0000000LL && db < 0xC0000000LL && d->H < 16000 && d->W < 16000 &&
        (linear || d->Cin >= whv2_min_cin)) {
#define YV4_WV_LAUNCH(LIN)                                                                                             \
  {                                                                                                                    \
    static LdsAttrOnce once_b, once_h;                                                                                 \
    if (int rc = ensure_dyn_lds(once_b, reinterpret_cast<const void*>(conv_wgrad_v2_h16_kernel<true, LIN>), (size_t)kWhLdsV2, "conv_wgrad_v2_h16")) return rc;  \
    if (int rc = ensure_dyn_lds(once_h, reinterpret_cast<const void*>(conv_wgrad_v2_h16_kernel<false, LIN>), (size_t)kWhLdsV2, "conv_wgrad_v2_h16")) return rc; \
    if (dtype == YV4_BF16) hipLaunchKernelGGL((conv_wgrad_v2_h16_kernel<true, LIN>), grid, dim3(256), (size_t)kWhLdsV2, hs, a, (unsigned)xb, (unsigned)db);     \
    else hipLaunchKernelGGL((conv_wgrad_v2_h16_kernel<false, LIN>), grid, dim3(256), (size_t)kWhLdsV2, hs, a, (unsigned)xb, (unsigned)db);                      \
  }
      if (linear) YV4_WV_LAUNCH(true)
      else YV4_WV_LAUNCH(false)
#undef YV4_WV_LAUNCH
      YV4_CHECK_LAUNCH("conv_wgrad_v2_h16");
      return finish();
    }
#define YV4_WH_LAUNCH(NB)                                                                                              \
  {                                                                                                                    \
    static LdsAttrOnce once_b, once_h;                                                                                 \
    if (int rc = ensure_dyn_lds(once_b, reinterpret_cast<const void*>(conv_wgrad_h16_kernel<true, NB>), ldsh, "conv_wgrad_h16")) return rc;  \
    if (int rc = ensure_dyn_lds(once_h, reinterpret_cast<const void*>(conv_wgrad_h16_kernel<false, NB>), ldsh, "conv_wgrad_h16")) return rc; \
    if (dtype == YV4_BF16) hipLaunchKernelGGL((conv_wgrad_h16_kernel<true, NB>), grid, dim3(256), ldsh, hs, a, (unsigned)xb, (unsigned)db);  \
    else hipLaunchKernelGGL((conv_wgrad_h16_kernel<false, NB>), grid, dim3(256), ldsh, hs, a, (unsigned)xb, (unsigned)db);                  \
  }
#ifdef YV4_MEASURE
    if (nbuf == 3) YV4_WH_LAUNCH(3)
    else if (nbuf == 4) YV4_WH_LAUNCH(4)
    else if (nbuf == 5) YV4_WH_LAUNCH(5)
    else
#endif
    YV4_WH_LAUNCH(2)
#undef YV4_WH_LAUNCH
    YV4_CHECK_LAUNCH("conv_wgrad_h16");
    return finish();
  }
  a.tiles_k = (a.K + 63) / 64;
  const int tiles_c = (a.Cout + 63) / 64;
  const long long tiles = (long long)a.tiles_k * tiles_c;
  const long long chunks = ch;
  const size_t lds = (size_t)4 * kWgRows * 64 * es;
  YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(conv_wgrad_kernel<T>, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), lds,
                                           reinterpret_cast<hipStream_t>(stream), a, (unsigned)xb, (unsigned)db));
  YV4_CHECK_LAUNCH("conv_wgrad");
  return finish();
}

// phase: 0 = sums + finalize (one rank), 1 = sums only (SyncBN: the caller all-reduces `work`), 2 = sums only, in the
// layout of a conv epilogue's replica 0 (fixed-point words stay words)
static int bn_stats_impl(int dtype, const void* x, int64_t M, int C, int x_cstride, int x_coff, float eps, float momentum,
                         double* work, float* mean, float* invstd, float* running_mean, float* running_var,
                         void* stream, int phase = 0) {
  YV4_REQUIRE(x && work && (phase != 0 || (mean && invstd)) && M > 0 && C > 0, "bn_train_stats: bad argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "bn_train_stats: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(C % 4 == 0 && x_cstride % 4 == 0 && x_coff % 4 == 0, "bn_train_stats: channels must be multiples of 4");
  YV4_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_stats: running stats come together");
  YV4_REQUIRE(C <= 4096, "bn_train_stats: more than 4096 channels");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int det = deterministic() ? 1 : 0;      // work: [hi (2*C) | lo (2*C)] fixed-point words
  if (det && C > 2048) {                        // 4 C doubles of LDS per workgroup: 64 KB at 2 048 channels
    set_error("bn_train_stats: deterministic mode takes at most 2048 channels (%d given)", C);
    return YV4_E_UNSUPPORTED;
  }
  if (hipMemsetAsync(work, 0, sizeof(double) * (det ? 4 : 2) * C, s) != hipSuccess) { set_error("bn_train_stats: memset failed"); return YV4_E_LAUNCH; }
  const int rpb = bn_rows_per_block(M);
  dim3 grid((unsigned)((M + rpb - 1) / rpb));
  YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(bn_stats_kernel<T>, grid, dim3(256), sizeof(double) * (det ? 4 : 2) * C, s,
                                           reinterpret_cast<const T*>(x), M, C, x_cstride, x_coff, work, rpb, det));
  if (phase == 0)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, s, work, M, C, eps, momentum, mean, invstd,
                       running_mean, running_var, (const double*)nullptr, det ? 2 : 1, 0, (double*)nullptr, det);
  else if (det && phase == 1)     // the caller (SyncBN) all-reduces doubles
    hipLaunchKernelGGL(fx_decode_kernel<kFxStat>, dim3((2 * C + 255) / 256), dim3(256), 0, s, work, 2 * C);
  YV4_CHECK_LAUNCH("bn_train_stats");
  return YV4_OK;
}

static int bn_fwd_impl(int dtype, const void* x, int x_cstride, int x_coff, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, const void* residual, int r_cstride, int r_coff, void* y,
                       int y_cstride, int y_coff, int64_t M, int C, int act, float slope, void* stream) {
  YV4_REQUIRE(x && mean && invstd && gamma && beta && y && M > 0 && C > 0, "bn_act_fwd: bad argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "bn_act_fwd: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(((C | x_cstride | x_coff | y_cstride | y_coff) & 3) == 0, "bn_act_fwd: channels must be multiples of 4");
  YV4_REQUIRE(!residual || ((r_cstride | r_coff) & 3) == 0, "bn_act_fwd: residual channels must be multiples of 4");
  BnArgs a = {};
  a.x = x; a.x_cs = x_cstride; a.x_co = x_coff; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
  a.res = residual; a.r_cs = r_cstride; a.r_co = r_coff; a.y = y; a.y_cs = y_cstride; a.y_co = y_coff;
  a.M = M; a.C = C; a.act = act; a.slope = slope;
  YV4_REQUIRE(C <= 4096, "bn_act_fwd: more than 4096 channels");
  a.rows_per_block = bn_rows_per_block(M);
  const dim3 grid((unsigned)((M + a.rows_per_block - 1) / a.rows_per_block));
  const bool v8 = dtype != YV4_F32 && g_bn_vec8 && ((C | x_cstride | x_coff | y_cstride | y_coff) & 7) == 0 &&
                  (!residual || ((r_cstride | r_coff) & 7) == 0);
  if (dtype != YV4_F32 && act == YV4_ACT_MISH && g_bn16) {
    const bool w8 = g_bn16_v == 8 && ((C | x_cstride | x_coff | y_cstride | y_coff) & 7) == 0 &&
                    (!residual || ((r_cstride | r_coff) & 7) == 0);
    YV4_DISPATCH_H16V(dtype, w8, hipLaunchKernelGGL((bn16_fwd_kernel<T, V, YV4_BN16_U>), grid, dim3(256), 0,
                                                    reinterpret_cast<hipStream_t>(stream), a));
    YV4_CHECK_LAUNCH("bn_act_fwd");
    return YV4_OK;
  }
  YV4_DISPATCH_TV(dtype, v8, hipLaunchKernelGGL((bn_act_fwd_kernel<T, V>), grid, dim3(256), 0,
                                                reinterpret_cast<hipStream_t>(stream), a));
  YV4_CHECK_LAUNCH("bn_act_fwd");
  return YV4_OK;
}

static int bn_bwd_impl(int dtype, const void* x, int x_cstride, int x_coff, const void* dy, int dy_cstride, int dy_coff,
                       const float* mean, const float* invstd, const float* gamma, const float* beta, void* dx,
                       int dx_cstride, int dx_coff, float* dgamma, float* dbeta, double* work, int64_t M, int C, int act,
                       float slope, void* stream, int eval_mode = 0, int phase = 0, int64_t M_total = 0,
                       const double* rows_dev = nullptr, int accumulate = 0, int work_is_zero = 0) {
  // phase 0: reduce + apply; 1: reduce only, dgamma / dbeta published from the LOCAL sums (SyncBN: the caller
  // then all-reduces `work`); 2: apply only, `work` holding the sums over M_total rows
  YV4_REQUIRE(x && dy && mean && invstd && gamma && beta && work && M > 0 && C > 0, "bn_act_bwd: bad argument");
  YV4_REQUIRE(phase == 2 || (dgamma && dbeta), "bn_act_bwd: dgamma / dbeta missing");
  YV4_REQUIRE(phase == 1 || dx, "bn_act_bwd: dx missing");
  YV4_REQUIRE(phase != 2 || rows_dev || M_total >= M, "bn_act_bwd: total row count below the local one");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "bn_act_bwd: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(((C | x_cstride | x_coff | dy_cstride | dy_coff | dx_cstride | dx_coff) & 3) == 0,
              "bn_act_bwd: channels must be multiples of 4");
  YV4_REQUIRE(C <= 4096, "bn_act_bwd: more than 4096 channels");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // deterministic mode: `work` is [hi (2*C) | lo (2*C)] fixed-point words between the reduction and the apply pass of
  // ONE call; what leaves the library (phase 1) or enters it (phase 2) is doubles
  const int det = deterministic() && phase != 2 ? 1 : 0;
  if (phase != 2 && !work_is_zero && hipMemsetAsync(work, 0, sizeof(double) * (det ? 4 : 2) * C, s) != hipSuccess) {
    set_error("bn_act_bwd: memset failed");
    return YV4_E_LAUNCH;
  }
  BnArgs a = {};
  a.x = x; a.x_cs = x_cstride; a.x_co = x_coff; a.dy = dy; a.dy_cs = dy_cstride; a.dy_co = dy_coff;
  a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.dx = dx; a.dx_cs = dx_cstride; a.dx_co = dx_coff;
  a.sums = work; a.M = M; a.C = C; a.act = act; a.slope = slope; a.eval_mode = eval_mode;
  a.dgamma = dgamma; a.dbeta = dbeta; a.det = det;
  a.M_total = phase == 2 ? M_total : M;
  a.publish = phase == 0 ? (accumulate ? 2 : 1) : 0;
  a.rows = phase == 2 ? rows_dev : nullptr;
  a.rows_per_block = bn_rows_per_block(M);
  dim3 grid((unsigned)((M + a.rows_per_block - 1) / a.rows_per_block));
  const bool v8 = dtype != YV4_F32 && g_bn_vec8 && ((C | x_cstride | x_coff | dy_cstride | dy_coff | dx_cstride | dx_coff) & 7) == 0;
  const bool b16 = dtype != YV4_F32 && act == YV4_ACT_MISH && g_bn16;
  const bool w8 = g_bn16_v == 8 && ((C | x_cstride | x_coff | dy_cstride | dy_coff | dx_cstride | dx_coff) & 7) == 0;
  if (phase != 2) {
    // channel groups of >= 64 channels (whole 128-byte lines of 16-bit rows), the row blocks shrunk so that the
    // workgroup count stays what bn_rows_per_block aims at
    static const int cg_min = YV4_ENV_INT("YV4_BN_RED_CG", 64);
    int groups = 1;
    if (cg_min > 0 && C % cg_min == 0 && C / cg_min >= 2) groups = C / cg_min < 16 ? C / cg_min : 16;
    while (groups > 1 && (C % groups != 0 || (C / groups) % 8 != 0)) --groups;
    BnArgs r = a;
    r.red_cg = C / groups;
    if (det && r.red_cg > 2048) {                // 4 doubles of LDS per channel of a group: 64 KB at 2 048
      set_error("bn_act_bwd: deterministic mode takes at most 2048 channels per reduction group (%d)", r.red_cg);
      return YV4_E_UNSUPPORTED;
    }
    int64_t rpb = (int64_t)a.rows_per_block * groups;
    if (rpb > g_bn_rows_cap) rpb = g_bn_rows_cap;
    r.rows_per_block = (int)rpb;
    const dim3 rgrid((unsigned)((M + rpb - 1) / rpb), (unsigned)groups);
    if (b16) {
      YV4_DISPATCH_H16V(dtype, w8, hipLaunchKernelGGL((bn16_bwd_reduce_kernel<T, V, YV4_BN16_U>), rgrid, dim3(256),
                                                      sizeof(double) * (det ? 4 : 2) * r.red_cg, s, r));
    } else {
      YV4_DISPATCH_TV(dtype, v8, hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, V>), rgrid, dim3(256), sizeof(double) * (det ? 4 : 2) * r.red_cg, s, r));
    }
  }
  if (phase == 1) {
    if (det) hipLaunchKernelGGL(fx_decode_kernel<kFxGrad>, dim3((2 * C + 255) / 256), dim3(256), 0, s, work, 2 * C);
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, work, C, dbeta);
    hipLaunchKernelGGL(sums_to_float_kernel, dim3((C + 255) / 256), dim3(256), 0, s, work + C, C, dgamma);
  } else {
    if (b16) {
      YV4_DISPATCH_H16V(dtype, w8, hipLaunchKernelGGL((bn16_bwd_apply_kernel<T, V, YV4_BN16_U>), grid, dim3(256), 0, s, a));
    } else {
      YV4_DISPATCH_TV(dtype, v8, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, V>), grid, dim3(256), 0, s, a));
    }
  }
  YV4_CHECK_LAUNCH("bn_act_bwd");
  return YV4_OK;
}

extern "C" int yv4_conv_wgrad(const yv4_conv_desc* d, const float* x, const float* dy, float* dw, void* stream) {
  return wgrad_impl(d, YV4_F32, x, dy, dw, stream);
}
extern "C" int yv4_conv_wgrad_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* dy, float* dw,
                                  void* stream) {
  YV4_REQUIRE(dtype == YV4_F16 || dtype == YV4_BF16, "wgrad_h16: dtype must be YV4_F16 or YV4_BF16");
  return wgrad_impl(d, dtype, x, dy, dw, stream);
}

// Deterministic weight gradient: the chunks of the M reduction store their partials to slabs of `workspace` and one
// small kernel adds them to dw in chunk order -- same accumulate-into-dw contract as yv4_conv_wgrad[_h16], run-to-run
// bit-identical, and the partial exchange moves at store speed instead of the ~1.3 TB/s of float atomics.
extern "C" size_t yv4_conv_wgrad_workspace(const yv4_conv_desc* d, int dtype) {
  if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  long long ch = 1, rw = 0;
  wgrad_chunks(d, dtype, &ch, &rw);
  if (ch <= 1) return 0;
  return (size_t)ch * (size_t)d->Cout * (size_t)(d->KH * d->KW * d->Cin) * sizeof(float);
}
extern "C" int yv4_conv_wgrad_det(const yv4_conv_desc* d, int dtype, const void* x, const void* dy, float* dw,
                                  float* workspace, size_t workspace_bytes, void* stream) {
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "wgrad_det: dtype must be f32, f16 or bf16");
  return wgrad_impl(d, dtype, x, dy, dw, stream, workspace, workspace_bytes);
}

extern "C" int yv4_dilate2_fwd(const float* src, float* dst, int N, int H, int W, int C, int src_cstride, int src_coff,
                               void* stream) {
  YV4_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0, "dilate2: bad argument");
  YV4_REQUIRE(C % 4 == 0 && src_cstride % 4 == 0 && src_coff % 4 == 0, "dilate2: channels must be multiples of 4");
  const size_t total = (size_t)N * 2 * H * 2 * W * (C / 4);
  hipLaunchKernelGGL(dilate2_kernel, dim3(ew_grid_t(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst,
                     N, H, W, C / 4, src_cstride, src_coff);
  YV4_CHECK_LAUNCH("dilate2");
  return YV4_OK;
}

extern "C" int yv4_bn_train_stats(const float* x, int64_t M, int C, int x_cstride, int x_coff, float eps, float momentum,
                                  double* work /* 4*C doubles */, float* mean, float* invstd, float* running_mean,
                                  float* running_var, void* stream) {
  return bn_stats_impl(YV4_F32, x, M, C, x_cstride, x_coff, eps, momentum, work, mean, invstd, running_mean, running_var,
                       stream);
}
extern "C" int yv4_bn_train_stats_h16(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff, float eps,
                                      float momentum, double* work, float* mean, float* invstd, float* running_mean,
                                      float* running_var, void* stream) {
  return bn_stats_impl(dtype, x, M, C, x_cstride, x_coff, eps, momentum, work, mean, invstd, running_mean, running_var,
                       stream);
}

extern "C" int yv4_bn_act_fwd(const float* x, int x_cstride, int x_coff, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* residual, int r_cstride, int r_coff,
                              float* y, int y_cstride, int y_coff, int64_t M, int C, int act, float slope, void* stream) {
  return bn_fwd_impl(YV4_F32, x, x_cstride, x_coff, mean, invstd, gamma, beta, residual, r_cstride, r_coff, y, y_cstride,
                     y_coff, M, C, act, slope, stream);
}
extern "C" int yv4_bn_act_fwd_h16(const void* x, int dtype, int x_cstride, int x_coff, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, const void* residual,
                                  int r_cstride, int r_coff, void* y, int y_cstride, int y_coff, int64_t M, int C, int act,
                                  float slope, void* stream) {
  return bn_fwd_impl(dtype, x, x_cstride, x_coff, mean, invstd, gamma, beta, residual, r_cstride, r_coff, y, y_cstride,
                     y_coff, M, C, act, slope, stream);
}

extern "C" int yv4_bn_act_bwd(const float* x, int x_cstride, int x_coff, const float* dy, int dy_cstride, int dy_coff,
                              const float* mean, const float* invstd, const float* gamma, const float* beta,
                              float* dx, int dx_cstride, int dx_coff, float* dgamma, float* dbeta,
                              double* work /* 4*C doubles */, int64_t M, int C, int act, float slope, void* stream) {
  return bn_bwd_impl(YV4_F32, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, dx, dx_cstride,
                     dx_coff, dgamma, dbeta, work, M, C, act, slope, stream);
}
extern "C" int yv4_bn_act_bwd_h16(const void* x, int dtype, int x_cstride, int x_coff, const void* dy, int dy_cstride,
                                  int dy_coff, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, void* dx, int dx_cstride, int dx_coff, float* dgamma, float* dbeta,
                                  double* work, int64_t M, int C, int act, float slope, void* stream) {
  return bn_bwd_impl(dtype, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, dx, dx_cstride,
                     dx_coff, dgamma, dbeta, work, M, C, act, slope, stream);
}

// yv4_conv_fwd_stats' fallback: the sums of y into the first replica (pair) of a cleared statistics buffer, left in the
// form yv4_bn_finalize(replicas = YV4_STATS_REPLICAS) reads
int bn_partial_sums_replica0(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff, double* stats,
                             void* stream) {
  return bn_stats_impl(dtype, x, M, C, x_cstride, x_coff, 0.f, 0.f, stats, nullptr, nullptr, nullptr, nullptr, stream, 2);
}

// ---- SyncBN: the same kernels with the cross-rank exchange between their two halves -----------------
extern "C" int yv4_bn_partial_sums(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff, double* work,
                                   void* stream) {
  return bn_stats_impl(dtype, x, M, C, x_cstride, x_coff, 0.f, 0.f, work, nullptr, nullptr, nullptr, nullptr, stream, 1);
}
extern "C" int yv4_bn_finalize(double* work, int replicas, int64_t M_total, const double* rows_dev, int C, float eps,
                               float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                               int clear_work, double* zero_after, void* stream) {
  YV4_REQUIRE(work && mean && invstd && (rows_dev || M_total > 0) && C > 0 && replicas >= 1, "bn_finalize: bad argument");
  YV4_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats come together");
  // replicas == YV4_STATS_REPLICAS: the buffer a conv epilogue filled (yv4_conv_fwd_stats) -- fixed-point replica pairs
  // in deterministic mode; any other count: plain doubles (SyncBN's all-reduced totals)
  const int det = deterministic() && replicas == YV4_STATS_REPLICAS ? 1 : 0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), work,
                     M_total, C, eps, momentum, mean, invstd, running_mean, running_var, rows_dev, replicas, clear_work ? 1 : 0,
                     zero_after, det);
  YV4_CHECK_LAUNCH("bn_finalize");
  return YV4_OK;
}
extern "C" int yv4_conv_stats_fold(double* stats, int C, int clear_stats, double* out, void* stream) {
  YV4_REQUIRE(stats && out && C > 0, "conv_stats_fold: bad argument");
  hipLaunchKernelGGL(stats_fold_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), stats,
                     C, YV4_STATS_REPLICAS, clear_stats ? 1 : 0, out, deterministic() ? 1 : 0);
  YV4_CHECK_LAUNCH("conv_stats_fold");
  return YV4_OK;
}
extern "C" int yv4_bn_act_bwd_sums(const void* x, int dtype, int x_cstride, int x_coff, const void* dy, int dy_cstride,
                                   int dy_coff, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* dgamma, float* dbeta, double* work, int64_t M, int C,
                                   int act, float slope, void* stream) {
  return bn_bwd_impl(dtype, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, nullptr, 4, 0,
                     dgamma, dbeta, work, M, C, act, slope, stream, 0, 1);
}
extern "C" int yv4_bn_act_bwd_apply(const void* x, int dtype, int x_cstride, int x_coff, const void* dy, int dy_cstride,
                                    int dy_coff, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, void* dx, int dx_cstride, int dx_coff, const double* work,
                                    int64_t M, int64_t M_total, const double* rows_dev, int C, int act, float slope,
                                    void* stream) {
  return bn_bwd_impl(dtype, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, dx, dx_cstride,
                     dx_coff, nullptr, nullptr, const_cast<double*>(work), M, C, act, slope, stream, 0, 2, M_total,
                     rows_dev);
}

// As yv4_bn_act_bwd_h16 / yv4_bn_eval_act_bwd, but dgamma / dbeta are ADDED to (the parameters' own .grad: no temporary,
// no accumulation kernel afterwards)
extern "C" int yv4_bn_act_bwd_accum(const void* x, int dtype, int x_cstride, int x_coff, const void* dy, int dy_cstride,
                                    int dy_coff, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, void* dx, int dx_cstride, int dx_coff, float* dgamma, float* dbeta,
                                    double* work, int64_t M, int C, int act, float slope, int flags, void* stream) {
  // flags: bit 0 = eval-mode BN, bit 1 = `work` is already zero (yv4_bn_finalize's zero_after cleared it)
  return bn_bwd_impl(dtype, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, dx, dx_cstride,
                     dx_coff, dgamma, dbeta, work, M, C, act, slope, stream, flags & 1, 0, 0, nullptr, 1, (flags >> 1) & 1);
}

extern "C" int yv4_bn_eval_act_bwd(const void* x, int dtype, int x_cstride, int x_coff, const void* dy, int dy_cstride,
                                   int dy_coff, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, void* dx, int dx_cstride, int dx_coff, float* dgamma, float* dbeta,
                                   double* work, int64_t M, int C, int act, float slope, void* stream) {
  return bn_bwd_impl(dtype, x, x_cstride, x_coff, dy, dy_cstride, dy_coff, mean, invstd, gamma, beta, dx, dx_cstride,
                     dx_coff, dgamma, dbeta, work, M, C, act, slope, stream, 1);
}

extern "C" int yv4_pack_weight(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw, int Cout, int Cin,
                               int KH, int KW, int KHo, int KWo, int kh0, int kh_step, int kw0, int kw_step, int transpose,
                               int pad_to, void* dst, int dtype, void* stream) {
  YV4_REQUIRE(w && dst && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && KHo > 0 && KWo > 0 && pad_to > 0,
              "pack_weight: bad argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "pack_weight: dtype must be f32, f16 or bf16");
  const int khl = kh0 + (KHo - 1) * kh_step, kwl = kw0 + (KWo - 1) * kw_step;
  YV4_REQUIRE(kh0 >= 0 && kh0 < KH && khl >= 0 && khl < KH && kw0 >= 0 && kw0 < KW && kwl >= 0 && kwl < KW,
              "pack_weight: tap selection leaves the %dx%d kernel", KH, KW);
  const int IC = transpose ? Cout : Cin, R = transpose ? Cin : Cout;
  const int ICp = (IC + pad_to - 1) / pad_to * pad_to;
  const long long nrows = (long long)R * KHo * KWo;
  YV4_REQUIRE(nrows < (1LL << 31), "pack_weight: too many rows");
  const unsigned grid = (unsigned)(nrows < 8192 ? nrows : 8192);
  YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(grid), dim3(256), 0,
                                           reinterpret_cast<hipStream_t>(stream), w, (long long)s_co, (long long)s_ci,
                                           (long long)s_kh, (long long)s_kw, Cout, Cin, KHo, KWo, kh0, kh_step, kw0, kw_step,
                                           transpose ? 1 : 0, ICp, reinterpret_cast<T*>(dst), (int)nrows));
  YV4_CHECK_LAUNCH("pack_weight");
  return YV4_OK;
}

extern "C" int yv4_pack_weights_multi(const yv4_pack_desc* table_dev, int n, int total_blocks, void* stream) {
  YV4_REQUIRE(table_dev && n > 0 && total_blocks > 0, "pack_weights_multi: bad argument");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     table_dev, n);
  YV4_CHECK_LAUNCH("pack_weights_multi");
  return YV4_OK;
}

extern "C" int yv4_spp_pool_bwd(const void* xcat, int x_cstride, int x_coff, const void* dcat, int d_cstride, int d_coff,
                                float* dx, int N, int H, int W, int C, int dtype, void* stream) {
  YV4_REQUIRE(xcat && dcat && dx && N > 0 && H > 0 && W > 0 && C > 0, "spp_pool_bwd: bad argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "spp_pool_bwd: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(((C | x_cstride | x_coff | d_cstride | d_coff) & 3) == 0, "spp_pool_bwd: channels must be multiples of 4");
  YV4_REQUIRE(x_coff + C <= x_cstride && d_coff + 4 * C <= d_cstride, "spp_pool_bwd: view exceeds its pixel stride");
  YV4_REQUIRE((long long)H * W < (1LL << 31), "spp_pool_bwd: H*W does not fit 31 bits");
  const bool f32 = dtype == YV4_F32;
  const int cg = f32 ? SppKey<float>::CG : SppKey<__bf16>::CG;
  const int det = deterministic() ? 1 : 0;
  const size_t lds = (size_t)H * W * cg * (3 * (f32 ? 8 : 4) + (det ? 8 : 4));
  if (lds <= 64 * 1024 && N <= 65535 && (long long)H * W * cg <= 256 * kSppItems) {   // small maps: keys and accumulator LDS-resident
    dim3 grid((unsigned)((C + cg - 1) / cg), (unsigned)N);
    YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(spp_pool_bwd_lds_kernel<T>, grid, dim3(256), lds,
                                             reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(xcat),
                                             x_cstride, x_coff, reinterpret_cast<const T*>(dcat), d_cstride, d_coff, dx, H,
                                             W, C, det));
    YV4_CHECK_LAUNCH("spp_pool_bwd");
    return YV4_OK;
  }
  if (det) {
    set_error("spp_pool_bwd: deterministic mode needs the LDS-resident form (H*W*%d <= %d, got %dx%d): the large-map "
              "kernel scatters with float atomics", cg, 256 * kSppItems, H, W);
    return YV4_E_UNSUPPORTED;
  }
  const size_t total = (size_t)N * H * W * (C / 4);
  YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(spp_pool_bwd_kernel<T>, dim3(ew_grid_t(total)), dim3(256), 0,
                                           reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(xcat),
                                           x_cstride, x_coff, reinterpret_cast<const T*>(dcat), d_cstride, d_coff, dx, N,
                                           H, W, C));
  YV4_CHECK_LAUNCH("spp_pool_bwd");
  return YV4_OK;
}


extern "C" int yv4_resample_nearest_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int Hd, int Wd, int C, int dy_cstride,
                                        int dy_coff, int dtype, void* stream) {
  YV4_REQUIRE(dy && dx && N > 0 && Hs > 0 && Ws > 0 && C > 0, "resample_bwd: bad argument");
  YV4_REQUIRE(dtype == YV4_F32 || dtype == YV4_F16 || dtype == YV4_BF16, "resample_bwd: dtype must be f32, f16 or bf16");
  YV4_REQUIRE(Hd % Hs == 0 && Wd % Ws == 0 && Hd / Hs <= 8 && Wd / Ws <= 8, "resample_bwd: integer scale factors up to 8 only");
  YV4_REQUIRE(((C | dy_cstride | dy_coff) & 3) == 0 && dy_coff >= 0 && dy_coff + C <= dy_cstride,
              "resample_bwd: channels must be multiples of 4 and the view inside its pixel stride");
  const size_t total = (size_t)N * Hs * Ws * (C / 4);
  YV4_DISPATCH_T(dtype, hipLaunchKernelGGL(resample_nearest_bwd_kernel<T>, dim3(ew_grid_t(total)), dim3(256), 0,
                                           reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const T*>(dy),
                                           reinterpret_cast<T*>(dx), N, Hs, Ws, Hd / Hs, Wd / Ws, C / 4, dy_cstride, dy_coff));
  YV4_CHECK_LAUNCH("resample_nearest_bwd");
  return YV4_OK;
}
