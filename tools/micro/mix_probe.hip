// Does v_fma_mixlo/mixhi_f16 + v_pk_add_f16 give the same bits as (half)(w * (float)v) + acc ?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ uint32_t mix_products(float w, uint32_t packed) {
	uint32_t p;
	asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=&v"(p) : "v"(w), "v"(packed));
	asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(p) : "v"(w), "v"(packed));
	return p;
}
__global__ void probe(uint32_t n, uint32_t* bad, uint32_t* first) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = i * 2654435761u + 99u;
	s = s * 1664525u + 1013904223u;
	float w = (float)(s >> 8) * (1.0f / 16777216.0f);
	if (i & 1) w *= 1e-3f;
	s = s * 1664525u + 1013904223u;
	uint32_t packed = s;
	// keep exponents moderate: clear the top exponent bit of both halves (no inf / nan)
	packed &= 0xBFFFBFFFu;
	if (i & 2) packed &= 0x8FFF8FFFu; // small magnitudes -> denormal products
	s = s * 1664525u + 1013904223u;
	uint32_t accp = s & 0xBFFFBFFFu;
	union { uint32_t u; half2_t h; half_t s[2]; } v, a, p, r_asm, r_c;
	v.u = packed; a.u = accp;
	p.u = mix_products(w, packed);
	r_asm.h = a.h + p.h;
	for (int k = 0; k < 2; ++k) {
		float prod = w * (float)v.s[k];
		asm volatile("" : "+v"(prod)); // keep the fp32 product a value of its own: no fusing into v_fma_mix
		r_c.s[k] = a.s[k] + (half_t)prod;
	}
	if (r_asm.u != r_c.u) {
		uint32_t slot = atomicAdd(bad, 1u);
		if (slot < 8) { first[slot * 4] = __float_as_uint(w); first[slot * 4 + 1] = packed; first[slot * 4 + 2] = r_asm.u; first[slot * 4 + 3] = r_c.u; }
	}
}
int main() {
	uint32_t *bad, *first;
	hipMalloc(&bad, 4); hipMalloc(&first, 128);
	hipMemset(bad, 0, 4); hipMemset(first, 0, 128);
	uint32_t n = 1u << 26;
	probe<<<n / 256, 256>>>(n, bad, first);
	uint32_t hb, hf[32];
	hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 128, hipMemcpyDeviceToHost);
	printf("mismatches %u of %u\n", hb, n);
	for (int k = 0; k < 8 && k < (int)hb; ++k) printf("  w=%08x packed=%08x asm=%08x c=%08x\n", hf[4 * k], hf[4 * k + 1], hf[4 * k + 2], hf[4 * k + 3]);
	return 0;
}
