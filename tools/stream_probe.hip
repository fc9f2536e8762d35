// Read-bandwidth probe: what does the k_scan access pattern (one wavefront per
// workgroup, each streaming its own contiguous range backwards, U loads of
// W bytes/lane in flight) achieve on this GPU?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T, int U>
__global__ void __launch_bounds__(64) k_stream(const T *__restrict__ p, size_t per_wave, unsigned long long *out)
{
	const size_t base = (size_t)blockIdx.x * per_wave;
	const unsigned lane = threadIdx.x;
	unsigned long long acc = 0;
	for (size_t i = per_wave; i >= (size_t)64 * U; i -= (size_t)64 * U) {
		T v[U];
#pragma unroll
		for (int u = 0; u < U; u++) v[u] = p[base + i - 64 * (u + 1) + lane];
#pragma unroll
		for (int u = 0; u < U; u++) {
			const unsigned *w = (const unsigned *)&v[u];
			for (unsigned j = 0; j < sizeof(T) / 4; j++) acc += w[j];
		}
	}
	if (acc == 0x1234567) out[0] = acc;
}

template <typename T, int U>
static void run(const void *d, size_t bytes, int waves, unsigned long long *d_out, const char *name)
{
	const size_t n = bytes / sizeof(T);
	const size_t per_wave = n / waves / (64 * U) * (64 * U);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k_stream<T, U>), dim3(waves), dim3(64), 0, 0, (const T *)d, per_wave, d_out);
	hipEventRecord(e0);
	for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_stream<T, U>), dim3(waves), dim3(64), 0, 0, (const T *)d, per_wave, d_out);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("%-28s waves %6d  %8.1f GB/s\n", name, waves, 5.0 * per_wave * waves * sizeof(T) / ms / 1e6);
}

int main()
{
	const size_t bytes = (size_t)4 << 30;
	void *d; unsigned long long *d_out;
	hipMalloc(&d, bytes); hipMalloc(&d_out, 8);
	hipMemset(d, 1, bytes);
	for (int waves : {4096, 16384, 65536}) {
		run<uint2, 1>(d, bytes, waves, d_out, "8B/lane x1 in flight");
		run<uint2, 4>(d, bytes, waves, d_out, "8B/lane x4 in flight");
		run<uint2, 8>(d, bytes, waves, d_out, "8B/lane x8 in flight");
		run<uint4, 2>(d, bytes, waves, d_out, "16B/lane x2 in flight");
		run<uint4, 4>(d, bytes, waves, d_out, "16B/lane x4 in flight");
		run<uint4, 8>(d, bytes, waves, d_out, "16B/lane x8 in flight");
	}
	return 0;
}
