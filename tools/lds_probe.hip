// LDS accumulate probe: what does one visit of the tile path cost as
//  (a) ds_read_b32 -> wait -> v_add -> ds_write_b32   (the current k_scan8 form)
//  (b) ds_add_f32, no return, no wait
// with the launch shape of k_scan8 (one wavefront per workgroup, ~10 KB LDS, so
// 16 waves per CU)?  Lanes hit pseudo-random docs of a 2048-doc tile.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_probe lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define TILE 2048

template <int MODE, int PAD>
__global__ void __launch_bounds__(64) k_probe(float *out, int iters, unsigned active)
{
	__shared__ float acc[TILE + 64 + PAD];
	const unsigned lane = threadIdx.x;
	for (unsigned i = lane; i < TILE + 64; i += 64) acc[i] = 0.0f;
	__builtin_amdgcn_wave_barrier();
	unsigned s = lane * 2654435761u + blockIdx.x * 40503u + 1;
	const float v = 1.0f + lane * 0.001f;
	for (int it = 0; it < iters; it++) {
		s = s * 1664525u + 1013904223u;
		const bool on = lane < active;
		const unsigned d = on ? (s >> 21) : TILE + lane;
		const float x = on ? v : 0.0f;
		if (MODE == 0) {
			const float a = acc[d];
			acc[d] = a + x;
		} else if (MODE == 1) {
			__hip_atomic_fetch_add(&acc[d], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else if (MODE == 2) {
			/* exec-masked atomic: only the active lanes take part */
			if (on) __hip_atomic_fetch_add(&acc[d], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else if (MODE == 3) {
			/* integer atomic OR on a bitmap word, no return */
			unsigned *bm = (unsigned *)acc;
			__hip_atomic_fetch_or(&bm[d >> 5], 1u << (d & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else if (MODE == 4) {
			/* byte read-or-write */
			unsigned char *bm = (unsigned char *)acc;
			const unsigned char o = bm[d];
			bm[d] = o | (unsigned char)(1u << (it & 7));
		} else if (MODE == 5) {
			/* integer atomic add, no return */
			unsigned *bm = (unsigned *)acc;
			__hip_atomic_fetch_add(&bm[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		}
		__builtin_amdgcn_wave_barrier();
	}
	__builtin_amdgcn_wave_barrier();
	float r = 0;
	for (unsigned i = lane; i < TILE; i += 64) r += acc[i];
	if (r == 12345.678f) out[0] = r;
}

template <int MODE, int PAD>
static void run(const char *name, int waves, int iters, unsigned active, float *d_out)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_probe<MODE, PAD>), dim3(waves), dim3(64), 0, 0, d_out, iters, active);
	hipEventRecord(e0);
	for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_probe<MODE, PAD>), dim3(waves), dim3(64), 0, 0, d_out, iters, active);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	ms /= 3;
	printf("%-34s active %2u  %7.3f ms  %6.2f G visits/s  %6.1f ns/visit/wave\n", name, active, ms,
	    (double)waves * iters / ms / 1e6, ms * 1e6 / iters / ((double)waves / (256 * 16) < 1 ? 1 : (double)waves / (256 * 16)));
}

int main()
{
	float *d_out; hipMalloc(&d_out, 64);
	const int waves = 65536, iters = 2000;
	for (unsigned active : {64u, 16u, 4u}) {
		run<0, 0>("read-add-write, 16 waves/CU", waves, iters, active, d_out);
		run<1, 0>("ds_add_f32 (dummy slots), 16/CU", waves, iters, active, d_out);
		run<2, 0>("ds_add_f32 (exec-masked), 16/CU", waves, iters, active, d_out);
		run<3, 0>("ds_or_b32 bitmap, 16/CU", waves, iters, active, d_out);
		run<5, 0>("ds_add_u32, 16/CU", waves, iters, active, d_out);
		run<4, 0>("byte read-or-write, 16/CU", waves, iters, active, d_out);
	}
	return 0;
}
