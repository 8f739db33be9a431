// How much LDS may a 256-thread workgroup use and still have three (four) of its kind resident on a gfx950 CU? (allocation granularity of
// the 160 KB). Build: hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy.hip -o tools/micro/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char dyn[];
__global__ __launch_bounds__(256) void k(int* out) {
	dyn[threadIdx.x] = (char)threadIdx.x;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = dyn[5];
}
int main() {
	int prev = -1;
	for (int bytes = 32768; bytes <= 65536; bytes += 64) {
		int n = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, (size_t)bytes) != hipSuccess) { printf("query failed at %d\n", bytes); return 1; }
		if (n != prev) { printf("%d bytes of LDS per workgroup: %d workgroups per CU\n", bytes, n); prev = n; }
	}
	return 0;
}
