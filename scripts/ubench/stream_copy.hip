// Streaming ceiling of one MI355X for the shapes the rigid warp uses.
//   hipcc --offload-arch=gfx950 -O3 stream_copy.hip -o stream_copy && ./stream_copy [name-filter]
// Every test moves a 40 x 4096 x 4096 fp32 stack (2.68 GB in, 2.68 GB out where it writes):
//   read_f4      global_load_dwordx4, xor-reduced, nothing written
//   fill_f4      global_store_dwordx4 only
//   copy_f4      float4 global -> global, grid-stride, U loads in flight per lane, plain / nt
//   tile_*       the warp's shape: a workgroup of 8 waves owns a 512 x 32 output tile; per frame a
//                (32 + 4) x (512 + 16) window goes HBM -> LDS by global_load_lds_dwordx4 (16 B per lane)
//                at a per-frame offset, two ds_read_b128 per window row and lane, one 16-byte store per
//                output row and lane.  "inblock": the 40 frames are a loop inside the workgroup (what the
//                fused sum needs); "fmajor": one workgroup per (tile, frame), frame-major dispatch.
// Times are HIP events around REPS back-to-back launches; run under rocprofv3 --kernel-trace --stats
// and --pmc FETCH_SIZE / WRITE_SIZE for the per-kernel figures kept in profiles/r03_stream_copy_*.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;

template <int U>
__global__ __launch_bounds__(256) void read_f4(const f4* __restrict__ in, int64_t n4, float* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += in[i];
  if (acc.x + acc.y + acc.z + acc.w == 1.2345e33f) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void fill_f4(f4* __restrict__ out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = v;
}

template <int U, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy_f4(const f4* __restrict__ in, f4* __restrict__ out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT_ST) __builtin_nontemporal_store(v[u], out + i + u * stride);
      else out[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) out[i] = in[i];
}

// block-contiguous copy: a workgroup owns CHUNK bytes at a time (rows of a frame rather than a
// stride over the whole buffer): same DRAM pages touched by one CU for longer
template <int U, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy_f4_blk(const f4* __restrict__ in, f4* __restrict__ out, int64_t n4) {
  const int64_t chunk = 256 * U;  // f4 per workgroup step
  for (int64_t base = (int64_t)blockIdx.x * chunk; base + chunk <= n4; base += (int64_t)gridDim.x * chunk) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f4* p = in + base + u * 256 + threadIdx.x;
      v[u] = NT_LD ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4* p = out + base + u * 256 + threadIdx.x;
      if (NT_ST) __builtin_nontemporal_store(v[u], p);
      else *p = v[u];
    }
  }
}

// ---------------------------------------------------------------- the warp's tile shape
struct TileArgs {
  const float* in;
  float* out;
  float* sum;
  int nframes, h, w;
  int tiles_x, tiles_y;
};

__device__ __forceinline__ void frame_offset(int f, int& sy, int& sx) {
  sy = (f * 7) % 15 - 7;
  sx = (f * 11) % 15 - 7;
}

// WX x WY waves, 8 output rows per wave.  HALO: window rows = tile rows + 4, window quads = tile quads + 4
// (as the 5-tap separable resampler needs); HALO = false: exactly the tile (a pure tiled copy).
template <int WX, int WY, int NBUF, bool INBLOCK, bool HALO, bool NT_LD, bool NT_ST, bool SUM, int SPREAD = 0>
__global__ __launch_bounds__(64 * WX * WY, NBUF == 1 ? 4 : 2) void tile_copy(TileArgs a) {
  constexpr int NWAVES = WX * WY;
  constexpr int TROWS = WY * 8 + (HALO ? 4 : 0);
  constexpr int QUADS = WX * 64 + (HALO ? 4 : 0);
  constexpr int NQ = TROWS * QUADS;
  constexpr int QUADS_PAD = ((NQ + 63) / 64) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* const b0 = reinterpret_cast<f4*>(smem);
  f4* const b1 = NBUF == 2 ? b0 + QUADS_PAD : b0;
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wvx = wave % WX, wvy = wave / WX;
  const int xt = txi * (256 * WX), yt = tyi * (8 * WY);
  const int x0 = xt + wvx * 256 + lane * 4, y0 = yt + wvy * 8;
  const int64_t hw = (int64_t)h * w;
  const int f_lo = INBLOCK ? 0 : blockIdx.y, f_hi = INBLOCK ? a.nframes : blockIdx.y + 1;
  float acc[8][4];
  if (SUM) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  }
  // SPREAD: workgroup `tile` walks the frames starting at frame (SPREAD * tile) % nframes -- what a set of
  // workgroups that have drifted apart in time looks like to the memory system
  auto fmap = [&](int f) { return SPREAD ? (f + SPREAD * tile) % a.nframes : f; };
  auto dma = [&](int fi, f4* dst) {
    const int f = fmap(fi);
    const float* fr = a.in + (int64_t)f * hw;
    int sy, sx;
    frame_offset(f, sy, sx);
    if (!HALO) { sy = 0; sx = 0; }
    const int ax = xt + sx - (HALO ? 1 : 0);
    for (int i = wave; i < QUADS_PAD / 64; i += NWAVES) {
      int q = i * 64 + lane;
      q = q < NQ ? q : NQ - 1;
      const int tr = q / QUADS, qc = q - tr * QUADS;
      int r = yt + sy - (HALO ? 1 : 0) + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = ax + 4 * qc;
      c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
      __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(dst + i * 64), 16, 0, NT_LD ? 2 : 0);
    }
  };
  const int strip = (wvy * 8) * QUADS + wvx * 64 + lane;
  dma(f_lo, b0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int f = f_lo; f < f_hi; ++f) {
    if (NBUF == 2 && f + 1 < f_hi) dma(f + 1, cur ? b0 : b1);
    const f4* t = (cur ? b1 : b0) + strip;
    float* orow = a.out + (int64_t)fmap(f) * hw + (int64_t)y0 * w + x0;
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) {
      f4 q0 = t[(ro + (HALO ? 2 : 0)) * QUADS];
      if (HALO) {
        const f4 q1 = t[(ro + 2) * QUADS + 1];
        q0 = q0 * 0.5f + q1 * 0.5f;
      }
      if (SUM) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[ro][k] += q0[k];
      }
      if (NT_ST) __builtin_nontemporal_store(q0, reinterpret_cast<f4*>(orow + (int64_t)ro * w));
      else *reinterpret_cast<f4*>(orow + (int64_t)ro * w) = q0;
    }
    if (f + 1 < f_hi) {
      if (NBUF == 1) {
        __syncthreads();
        dma(f + 1, b0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
        cur ^= 1;
      }
    }
  }
  if (SUM) {
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) {
      f4 v = {acc[ro][0], acc[ro][1], acc[ro][2], acc[ro][3]};
      *reinterpret_cast<f4*>(a.sum + (int64_t)(y0 + ro) * w + x0) = v;
    }
  }
}

// ---------------------------------------------------------------- paced in-block copy
// The same in-block tile copy (halo, sum, nt stores, single buffer, 512 x 32 tiles), PERSISTENT: the grid
// is one round of co-resident workgroups (2 per CU), workgroup b takes tiles b, b + G, ...  LAG > 0: a
// workgroup does not start frame f before `arrived[round][f - LAG]` says every workgroup of the round has
// finished frame f - LAG -- a bounded wait (pacing only: no data is exchanged, so no fences; when the
// budget runs out the workgroup goes on).  Measures what drift between workgroups costs.
template <int LAG>
__global__ __launch_bounds__(512, 4) void tile_paced(TileArgs a, unsigned* __restrict__ arrived, int spin_budget, int nap) {
  constexpr int WX = 2, WY = 4, NWAVES = 8, TROWS = 36, QUADS = 132, NQ = TROWS * QUADS;
  constexpr int QUADS_PAD = ((NQ + 63) / 64) * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f4* const b0 = reinterpret_cast<f4*>(smem);
  const int nt = a.tiles_x * a.tiles_y, G = gridDim.x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wvx = wave % WX, wvy = wave / WX;
  const int64_t hw = (int64_t)h * w;
  int round = 0;
  for (int b = blockIdx.x; b < nt; b += G, ++round) {
    int tile = b;
    if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
    const int xt = txi * (256 * WX), yt = tyi * (8 * WY);
    const int x0 = xt + wvx * 256 + lane * 4, y0 = yt + wvy * 8;
    const int ng = (nt - round * G) < G ? (nt - round * G) : G;  // workgroups in this round
    unsigned* arr = arrived + round * a.nframes;
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
    auto dma = [&](int f) {
      const float* fr = a.in + (int64_t)f * hw;
      int sy, sx;
      frame_offset(f, sy, sx);
      const int ax = xt + sx - 1;
      for (int i = wave; i < QUADS_PAD / 64; i += NWAVES) {
        int q = i * 64 + lane;
        q = q < NQ ? q : NQ - 1;
        const int tr = q / QUADS, qc = q - tr * QUADS;
        int r = yt + sy - 1 + tr;
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        int c = ax + 4 * qc;
        c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
        __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(b0 + i * 64), 16, 0, 0);
      }
    };
    const int strip = (wvy * 8) * QUADS + wvx * 64 + lane;
    __syncthreads();  // previous tile's readers are done with the window
    if (LAG == 0 && spin_budget > 0 && round == 0) {  // initial stagger: up to spin_budget x 2 us, hashed by tile
      const int naps = (int)(((unsigned)tile * 2654435761u) >> 16) % (unsigned)spin_budget;
      for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(63);
    }
    dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int f = 0; f < a.nframes; ++f) {
      const f4* t = b0 + strip;
      float* orow = a.out + (int64_t)f * hw + (int64_t)y0 * w + x0;
#pragma unroll
      for (int ro = 0; ro < 8; ++ro) {
        f4 q0 = t[(ro + 2) * QUADS];
        const f4 q1 = t[(ro + 2) * QUADS + 1];
        q0 = q0 * 0.5f + q1 * 0.5f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[ro][k] += q0[k];
        __builtin_nontemporal_store(q0, reinterpret_cast<f4*>(orow + (int64_t)ro * w));
      }
      if (f + 1 < a.nframes) {
        __syncthreads();
        if (LAG > 0 && threadIdx.x == 0) {
          __hip_atomic_fetch_add(&arr[f], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (f + 1 - LAG >= 0) {
            int budget = spin_budget;
            while (__hip_atomic_load(&arr[f + 1 - LAG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ng && --budget > 0)
              for (int z = 0; z < a.tiles_x /*reused: naps per poll*/ * 0 + nap; ++z) __builtin_amdgcn_s_sleep(63);
          }
        }
        if (LAG > 0) __syncthreads();
        dma(f + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
#pragma unroll
    for (int ro = 0; ro < 8; ++ro) {
      f4 v = {acc[ro][0], acc[ro][1], acc[ro][2], acc[ro][3]};
      *reinterpret_cast<f4*>(a.sum + (int64_t)(y0 + ro) * w + x0) = v;
    }
  }
}

static const char* g_filter = nullptr;
static int g_reps = 10;

template <class F>
static void run(const char* name, double bytes, F launch) {
  if (g_filter && !strstr(name, g_filter)) return;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f, tot = 0.f;
  for (int r = 0; r < g_reps; ++r) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    tot += ms;
  }
  CK(hipGetLastError());
  printf("%-44s avg %7.3f ms  best %7.3f ms  %7.1f GB/s avg  %7.1f GB/s best\n", name, tot / g_reps, best,
         bytes / (tot / g_reps) * 1e-6, bytes / best * 1e-6);
  fflush(stdout);
}

template <int WX, int WY, int NBUF, bool INBLOCK, bool HALO, bool NT_LD, bool NT_ST, bool SUM, int SPREAD = 0>
static void run_tile(const char* name, TileArgs a, double bytes) {
  constexpr int TROWS = WY * 8 + (HALO ? 4 : 0), QUADS = WX * 64 + (HALO ? 4 : 0);
  constexpr int QUADS_PAD = ((TROWS * QUADS + 63) / 64) * 64;
  const int lds = QUADS_PAD * 16 * NBUF;
  a.tiles_x = a.w / (256 * WX);
  a.tiles_y = a.h / (8 * WY);
  auto k = tile_copy<WX, WY, NBUF, INBLOCK, HALO, NT_LD, NT_ST, SUM, SPREAD>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  dim3 grid(a.tiles_x * a.tiles_y, INBLOCK ? 1 : a.nframes);
  run(name, bytes, [&] { hipLaunchKernelGGL(k, grid, dim3(64 * WX * WY), lds, 0, a); });
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "all")) g_filter = argv[1];
  if (argc > 2) g_reps = atoi(argv[2]);
  const int T = 40, H = 4096, W = 4096;
  const int64_t n = (int64_t)T * H * W, n4 = n / 4;
  float *in, *out, *sum, *sink;
  CK(hipMalloc(&in, n * 4));
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&sum, (int64_t)H * W * 4));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(in, 0x3c, n * 4));  // finite, non-trivial bit patterns
  CK(hipMemset(out, 0, n * 4));
  const double B = (double)n * 4;
  const f4* in4 = reinterpret_cast<const f4*>(in);
  f4* out4 = reinterpret_cast<f4*>(out);

  for (int bpc : {4, 8}) {
    char nm[96];
    const int g = 256 * bpc;
    snprintf(nm, sizeof nm, "read_f4 U4 %d blocks/CU", bpc);
    run(nm, B, [&] { hipLaunchKernelGGL(read_f4<4>, dim3(g), dim3(256), 0, 0, in4, n4, sink); });
    snprintf(nm, sizeof nm, "read_f4 U8 %d blocks/CU", bpc);
    run(nm, B, [&] { hipLaunchKernelGGL(read_f4<8>, dim3(g), dim3(256), 0, 0, in4, n4, sink); });
    snprintf(nm, sizeof nm, "fill_f4 %d blocks/CU", bpc);
    run(nm, B, [&] { hipLaunchKernelGGL(fill_f4, dim3(g), dim3(256), 0, 0, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U1 plain %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<1, false, false>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U4 plain %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<4, false, false>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U8 plain %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<8, false, false>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U4 nt-ld %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<4, true, false>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U4 nt-st %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<4, false, true>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4 U4 nt-ld nt-st %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4<4, true, true>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4_blk U4 plain %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4_blk<4, false, false>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
    snprintf(nm, sizeof nm, "copy_f4_blk U8 nt-ld nt-st %d blocks/CU", bpc);
    run(nm, 2 * B, [&] { hipLaunchKernelGGL((copy_f4_blk<8, true, true>), dim3(g), dim3(256), 0, 0, in4, out4, n4); });
  }
  // one block per 1024 float4 (no grid stride): what a naive launch does
  run("copy_f4 U1 plain one-shot grid", 2 * B,
      [&] { hipLaunchKernelGGL((copy_f4<1, false, false>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, in4, out4, n4); });

  TileArgs a{in, out, sum, T, H, W, 0, 0};
  //            WX WY NBUF INBLK HALO  NTLD   NTST   SUM
  run_tile<2, 4, 1, true, false, false, false, false>("tile 512x32 inblock nohalo", a, 2 * B);
  run_tile<2, 4, 1, true, true, false, false, false>("tile 512x32 inblock halo", a, 2 * B);
  run_tile<2, 4, 1, true, true, false, false, true>("tile 512x32 inblock halo +sum", a, 2 * B);
  run_tile<2, 4, 1, true, true, true, false, true>("tile 512x32 inblock halo +sum nt-ld", a, 2 * B);
  run_tile<2, 4, 1, true, true, false, true, true>("tile 512x32 inblock halo +sum nt-st", a, 2 * B);
  run_tile<2, 4, 1, true, true, true, true, true>("tile 512x32 inblock halo +sum nt-ld nt-st", a, 2 * B);
  run_tile<2, 4, 2, true, true, false, false, true>("tile 512x32 inblock halo +sum 2buf", a, 2 * B);
  run_tile<2, 4, 2, true, true, false, true, true>("tile 512x32 inblock halo +sum 2buf nt-st", a, 2 * B);
  run_tile<2, 4, 1, false, false, false, false, false>("tile 512x32 fmajor nohalo", a, 2 * B);
  run_tile<2, 4, 1, false, true, false, false, false>("tile 512x32 fmajor halo", a, 2 * B);
  run_tile<2, 4, 1, false, true, false, true, false>("tile 512x32 fmajor halo nt-st", a, 2 * B);
  run_tile<2, 4, 1, false, true, true, true, false>("tile 512x32 fmajor halo nt-ld nt-st", a, 2 * B);
  run_tile<2, 4, 1, true, true, false, true, true, 1>("tile 512x32 inblock halo +sum nt-st spread1", a, 2 * B);
  run_tile<2, 4, 1, true, true, false, true, true, 7>("tile 512x32 inblock halo +sum nt-st spread7", a, 2 * B);
  {
    unsigned* arrived;
    CK(hipMalloc(&arrived, 8 * T * sizeof(unsigned)));
    TileArgs p = a;
    p.tiles_x = W / 512;
    p.tiles_y = H / 32;
    const int lds = ((36 * 132 + 63) / 64) * 64 * 16;
    auto go = [&](const char* nm, auto k, int G, int budget, int nap = 1) {
      CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      run(nm, 2 * B, [&] {
        CK(hipMemsetAsync(arrived, 0, 8 * T * sizeof(unsigned), 0));
        hipLaunchKernelGGL(k, dim3(G), dim3(512), lds, 0, p, arrived, budget, nap);
      });
    };
    go("paced persistent G512 lag0 (no pacing)", tile_paced<0>, 512, 0);
    go("paced persistent G512 lag0 stagger<8us", tile_paced<0>, 512, 4);
    go("paced persistent G512 lag0 stagger<16us", tile_paced<0>, 512, 8);
    go("paced persistent G512 lag0 stagger<32us", tile_paced<0>, 512, 16);
    go("paced persistent G512 lag0 stagger<64us", tile_paced<0>, 512, 32);
    go("paced persistent G512 lag1 nap1", tile_paced<1>, 512, 2000, 1);
    go("paced persistent G512 lag2 nap1", tile_paced<2>, 512, 2000, 1);
    go("paced persistent G512 lag2 nap4", tile_paced<2>, 512, 500, 4);
    go("paced persistent G512 lag3 nap2", tile_paced<3>, 512, 1000, 2);
    go("paced persistent G512 lag4 nap2", tile_paced<4>, 512, 1000, 2);
    go("paced persistent G512 lag4 nap8", tile_paced<4>, 512, 250, 8);
    go("paced persistent G512 lag8 nap8", tile_paced<8>, 512, 250, 8);
  }
  run_tile<1, 4, 1, true, true, false, false, true>("tile 256x32 inblock halo +sum", a, 2 * B);
  run_tile<4, 2, 1, true, true, false, false, true>("tile 1024x16 inblock halo +sum", a, 2 * B);
  run_tile<2, 8, 1, true, true, false, false, true>("tile 512x64 inblock halo +sum (16 waves)", a, 2 * B);
  return 0;
}
