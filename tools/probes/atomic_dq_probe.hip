// Probe: what does accumulating dQ with fp32 global atomics cost, in the access pattern a fused dK/dV+dQ attention
// backward would have?  Grid = (kv blocks of 128, heads, batch); every wave adds a [32 q][32 d] fp32 block per 64-query
// tile (wave w: q half = w & 1, d half = w >> 1), 16 atomic instructions of 64 lanes each (two 128-byte rows).
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomic_dq_probe.hip -o atomic_dq_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void probe(float* dq, int Tq, int H, int rotate, int work) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int64_t bh = (int64_t)blockIdx.z * H + blockIdx.y;
  float* base = dq + bh * Tq * 64;
  const int ntiles = Tq / 64;
  float v = (float)lane;
  for (int t = 0; t < ntiles; ++t) {
    const int tile = rotate ? (t + blockIdx.x * 2) % ntiles : t;
    // some ALU work between the bursts, as the real kernel would have (work fmas per value)
    for (int i = 0; i < work; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    const int q0 = tile * 64 + (wave & 1) * 32, d0 = (wave >> 1) * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      unsafeAtomicAdd(base + (int64_t)(q0 + row) * 64 + d0 + c, v);
    }
  }
}

int main(int argc, char** argv) {
  const int B = 8, H = 12, Tq = 1536, KV = 12;
  float* dq;
  const size_t n = (size_t)B * H * Tq * 64;
  hipMalloc(&dq, n * 4);
  hipMemset(dq, 0, n * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rotate = 0; rotate < 2; ++rotate)
    for (int work : {0, 200, 1000}) {
      for (int it = 0; it < 3; ++it) probe<<<dim3(KV, H, B), 256>>>(dq, Tq, H, rotate, work);
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it) probe<<<dim3(KV, H, B), 256>>>(dq, Tq, H, rotate, work);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      printf("rotate %d  alu %4d fma/tile: %.1f us per launch (%.0f M atomics, %.1f G/s)\n", rotate, work, ms * 100,
             (double)KV * H * B * 4 * 24 * 16 * 64 / 1e6, (double)KV * H * B * 4 * 24 * 16 * 64 / (ms / 10 * 1e-3) / 1e9);
    }
  std::vector<float> hbuf(64);
  hipMemcpy(hbuf.data(), dq, 256, hipMemcpyDeviceToHost);
  printf("check %g\n", hbuf[5]);
  return 0;
}
