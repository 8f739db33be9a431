// Microbenchmark: how many scattered (per-lane random) loads per clock does one MI355X CU sustain, by access
// width and table size? Sizes the hash-grid gather ceiling of render_nerf_fused.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/micro/gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T> __device__ uint32_t fold(T v);
template <> __device__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ uint32_t fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// MODE 0: every lane its own random entry. MODE 1: lanes in groups of 4 read 4 consecutive entries (one 4x wider
// contiguous piece per group). MODE 2: pairs of lanes read 2 consecutive entries.
// lanes_on: how many of a wave's 64 lanes issue the loads (exec mask) -- is a gather priced per wave-instruction or per active lane?
// stride_sel 1: every (64 / lanes_on)-th lane; 0: the first lanes_on lanes
template <typename T, int UNROLL, int MODE>
__global__ __launch_bounds__(256) void gather(const T* __restrict__ table, uint32_t mask, int iters, uint32_t* out, uint32_t lanes_on = 64u, uint32_t stride_sel = 0u) {
	uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	uint32_t acc = 0;
	const uint32_t lane = threadIdx.x & 63u;
	const bool on = stride_sel ? (lane % (64u / lanes_on)) == 0u : lane < lanes_on;
	if (!on) return;
	for (int it = 0; it < iters; ++it) {
		T v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			s = s * 1664525u + 1013904223u;
			uint32_t r = s >> 4;
			if (MODE == 1) r = (__shfl(r, (int)(lane & ~3u), 64) & ~3u) | (lane & 3u);
			if (MODE == 2) r = (__shfl(r, (int)(lane & ~1u), 64) & ~1u) | (lane & 1u);
			if (MODE == 3) r = (__shfl(r, (int)(lane & ~16u), 64) & ~1u) | ((lane >> 4) & 1u);
			if (MODE == 4) r = (__shfl(r, (int)(lane & ~32u), 64) & ~1u) | ((lane >> 5) & 1u);
			if (MODE == 5) r = (__shfl(r, (int)(lane & ~48u), 64) & ~3u) | ((lane >> 4) & 3u);
			if (MODE == 6) r = (__shfl(r, (int)(lane & ~4u), 64) & ~1u) | ((lane >> 2) & 1u);
			v[u] = table[r & mask];
		}
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc ^= fold<T>(v[u]);
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// 16-byte loads whose address is 8 modulo 16 (every load straddles two aligned 16-byte slots; one in eight also straddles a 64-byte boundary)
// against aligned ones: what the pair gathers of tools/experiments would issue for dense levels with odd x
template <int UNROLL, int MISALIGN>
__global__ __launch_bounds__(256) void gather16_at(const char* __restrict__ table, uint32_t mask, int iters, uint32_t* out) {
	uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			s = s * 1664525u + 1013904223u;
			const uint32_t r = (s >> 4) & mask;
			v[u] = *(const uint4*)(table + (size_t)r * 16u + MISALIGN);
		}
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}
template <int MISALIGN>
void run16(void* d_table, size_t table_bytes, uint32_t* d_out) {
	const uint32_t mask = (uint32_t)(table_bytes / 16) - 2u; // (leave the last slot: the misaligned load reads 8 bytes past its own)
	const int blocks = 2048, iters = 500;
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	gather16_at<16, MISALIGN><<<blocks, 256>>>((const char*)d_table, mask & ~1u, iters / 4, d_out);
	CHECK(hipEventRecord(a));
	gather16_at<16, MISALIGN><<<blocks, 256>>>((const char*)d_table, mask & ~1u, iters, d_out);
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	const double loads = (double)blocks * 256.0 * iters * 16;
	printf("16 B random at offset %d mod 16      table %10.4f MB blocks %4d: %8.3f ms  %5.2f lane-loads/clk/CU (2.4 GHz)\n", MISALIGN, table_bytes / 1048576.0, blocks, ms, loads / (ms * 1e-3 * 2.4e9 * 256.0));
}

template <typename T, int UNROLL, int MODE>
void run(const char* name, void* d_table, size_t table_bytes, int blocks, uint32_t* d_out, uint32_t lanes_on = 64u, uint32_t stride_sel = 0u) {
	uint32_t n = (uint32_t)(table_bytes / sizeof(T));
	uint32_t mask = n - 1;
	int iters = 2000 / UNROLL * 4;
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a));
	CHECK(hipEventCreate(&b));
	gather<T, UNROLL, MODE><<<blocks, 256>>>((const T*)d_table, mask, iters / 4, d_out, lanes_on, stride_sel);
	CHECK(hipEventRecord(a));
	gather<T, UNROLL, MODE><<<blocks, 256>>>((const T*)d_table, mask, iters, d_out, lanes_on, stride_sel);
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	double loads = (double)blocks * 256.0 * iters * UNROLL * (lanes_on / 64.0);
	double per_clk_cu = loads / (ms * 1e-3 * 2.4e9 * 256.0);
	if (lanes_on != 64u) printf("[%2u of 64 lanes, %s] wave-instructions/clk/CU %.4f  ", lanes_on, stride_sel ? "strided" : "first", per_clk_cu / lanes_on);
	printf("%-34s table %10.4f MB blocks %4d: %8.3f ms  %7.1f Glane-loads/s  %5.2f lane-loads/clk/CU (2.4 GHz)  %6.0f GB/s useful\n", name,
	       table_bytes / 1048576.0, blocks, ms, loads / ms * 1e-6, per_clk_cu, loads * sizeof(T) / ms * 1e-6);
}

int main(int argc, char** argv) {
	size_t max_bytes = 1ull << 30;
	void* d_table;
	uint32_t* d_out;
	CHECK(hipMalloc(&d_table, max_bytes));
	CHECK(hipMemset(d_table, 1, max_bytes));
	CHECK(hipMalloc(&d_out, 4));
	if (argc > 1 && argv[1][0] == 'u') { // "unaligned": 16-byte loads at 8 modulo 16
		for (size_t sz : {8ull << 10, 16ull << 10, 1ull << 20, 32ull << 20}) {
			run16<0>(d_table, sz, d_out);
			run16<8>(d_table, sz, d_out);
		}
		return 0;
	}
	if (argc > 1) { // "masked": the same scattered 8-byte gather with 64 / 32 / 16 / 8 lanes of a wave switched on
		for (size_t sz : {8ull << 10, 1ull << 20, 32ull << 20}) {
			run<uint2, 16, 0>("8 B random, 16 in flight", d_table, sz, 2048, d_out);
			for (uint32_t on : {32u, 16u, 8u})
				for (uint32_t st : {0u, 1u}) run<uint2, 16, 0>("8 B random, 16 in flight", d_table, sz, 2048, d_out, on, st);
		}
		return 0;
	}
	size_t sizes[] = {256ull, 1ull << 10, 2ull << 10, 4ull << 10, 8ull << 10, 16ull << 10, 1ull << 20, 32ull << 20}; // 4 lines (every lane shares one), L1-resident, L2-resident, beyond L2
	for (size_t sz : sizes) {
		for (int blocks : {512, 2048}) {
			run<uint32_t, 16, 0>("4 B random, 16 in flight", d_table, sz, blocks, d_out);
			run<uint2, 16, 0>("8 B random, 16 in flight", d_table, sz, blocks, d_out);
			run<uint2, 32, 0>("8 B random, 32 in flight", d_table, sz, blocks, d_out);
			run<uint4, 16, 0>("16 B random, 16 in flight", d_table, sz, blocks, d_out);
			run<uint2, 16, 2>("8 B, lane pairs adjacent", d_table, sz, blocks, d_out);
			run<uint2, 16, 1>("8 B, lane quads adjacent (32 B)", d_table, sz, blocks, d_out);
			run<uint2, 16, 6>("8 B, pairs lane^4", d_table, sz, blocks, d_out);
			run<uint2, 16, 3>("8 B, pairs lane^16", d_table, sz, blocks, d_out);
			run<uint2, 16, 4>("8 B, pairs lane^32", d_table, sz, blocks, d_out);
			run<uint2, 16, 5>("8 B, quads lane^16^32", d_table, sz, blocks, d_out);
			run<uint4, 16, 1>("16 B, lane quads adjacent (64 B)", d_table, sz, blocks, d_out);
		}
	}
	return 0;
}
