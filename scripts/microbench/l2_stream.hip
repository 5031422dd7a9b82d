// How fast can ONE workgroup per CU stream an L2-resident buffer into LDS?  (the feature blocks of sage_bcm.hip: 26 KB per MFMA step)
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip && ./l2_stream
// Variants: plain loads + ds_write_b128 by W waves with D blocks in flight, or LDS DMA.  Reports bytes / clock / CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int PIECES_PER_WAVE, int DEPTH, bool DMA>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* __restrict__ src, int n_blocks, int block_bytes, int movers, int reps,
                                                     unsigned long long* cycles, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mover = wave < movers;
  u32x4 rb[DEPTH][PIECES_PER_WAVE];
  const unsigned long long t0 = __builtin_readcyclecounter();
  float acc = 0.f;
  int total = n_blocks * reps;
  if (mover) {
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int s = 0; s < PIECES_PER_WAVE; ++s) {
        const unsigned char* p = src + (size_t)(d % n_blocks) * block_bytes + (size_t)(wave + movers * s) * 1024 + lane * 16;
        if (DMA) __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(lds + (d % 3) * 32768 + (wave + movers * s) * 1024), 16, 0, 0);
        else rb[d][s] = *reinterpret_cast<const u32x4*>(p);
      }
  }
  for (int b = 0; b < total; b += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int blk = b + d;
      if (mover) {
        if (!DMA) {
#pragma unroll
          for (int s = 0; s < PIECES_PER_WAVE; ++s) *reinterpret_cast<u32x4*>(lds + (blk % 3) * 32768 + (wave + movers * s) * 1024 + lane * 16) = rb[d][s];
        } else {
          // wait for the oldest block: DEPTH - 1 younger blocks may stay in flight
          if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if ((DEPTH - 1) * PIECES_PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else if ((DEPTH - 1) * PIECES_PER_WAVE == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else if ((DEPTH - 1) * PIECES_PER_WAVE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else if ((DEPTH - 1) * PIECES_PER_WAVE == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int nb = (blk + DEPTH) % n_blocks;
#pragma unroll
        for (int s = 0; s < PIECES_PER_WAVE; ++s) {
          const unsigned char* p = src + (size_t)nb * block_bytes + (size_t)(wave + movers * s) * 1024 + lane * 16;
          if (DMA) __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(lds + ((blk + DEPTH) % 3) * 32768 + (wave + movers * s) * 1024), 16, 0, 0);
          else rb[d][s] = *reinterpret_cast<const u32x4*>(p);
        }
      }
      __syncthreads();
      acc += reinterpret_cast<float*>(lds)[(blk % 3) * 8192 + threadIdx.x];  // a consumer
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int PPW, int DEPTH, bool DMA>
void run(const char* name, const unsigned char* src, int n_blocks, int block_bytes, int movers, unsigned long long* cyc, float* sink) {
  const int reps = 40, grid = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<PPW, DEPTH, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((stream_kernel<PPW, DEPTH, DMA>), dim3(grid), dim3(512), 120 * 1024, 0, src, n_blocks, block_bytes, movers, reps, cyc, sink);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < grid; ++i) mean += (double)h[i];
  mean /= grid;
  const double bytes = (double)n_blocks * reps * (double)(movers * PPW * 1024);
  printf("%-44s movers=%d pieces/wave=%d depth=%d: %8.0f cycles/block  %6.2f B/clk/CU\n", name, movers, PPW, DEPTH, mean / (n_blocks * reps), bytes / mean);
}

int main() {
  const int block_bytes = 28 * 1024, n_blocks = 125;  // 3.5 MB: the fp32 feature planes of D = 400
  unsigned char* src;
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&src, (size_t)block_bytes * n_blocks);
  hipMemset(src, 1, (size_t)block_bytes * n_blocks);
  hipMalloc(&cyc, 256 * 8);
  hipMalloc(&sink, 256 * 512 * 4);
  run<7, 1, false>("loads + ds_write, 4 movers", src, n_blocks, block_bytes, 4, cyc, sink);
  run<7, 2, false>("loads + ds_write, 4 movers", src, n_blocks, block_bytes, 4, cyc, sink);
  run<4, 1, false>("loads + ds_write, 7 movers", src, n_blocks, block_bytes, 7, cyc, sink);
  run<4, 2, false>("loads + ds_write, 7 movers", src, n_blocks, block_bytes, 7, cyc, sink);
  run<4, 3, false>("loads + ds_write, 7 movers", src, n_blocks, block_bytes, 7, cyc, sink);
  run<7, 1, true>("LDS DMA, 4 movers", src, n_blocks, block_bytes, 4, cyc, sink);
  run<7, 2, true>("LDS DMA, 4 movers", src, n_blocks, block_bytes, 4, cyc, sink);
  run<7, 3, true>("LDS DMA, 4 movers", src, n_blocks, block_bytes, 4, cyc, sink);
  run<4, 2, true>("LDS DMA, 7 movers", src, n_blocks, block_bytes, 7, cyc, sink);
  run<4, 3, true>("LDS DMA, 7 movers", src, n_blocks, block_bytes, 7, cyc, sink);
  return 0;
}
