// Issue rate of vector instructions on gfx950 (the instructions the seed scan, the top-k and the FP64 kernels are made of):
// W waves per SIMD, each a stream of independent instructions on 16 accumulators.  Prints cycles per wave-instruction per SIMD
// at the nominal clock.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
/* 32-bit forms: %0 accumulator, %1 / %2 vector operands, s4 a scalar operand */
#define OPS32(X) \
	X(0, "v_xor_b32 %0, %1, %0") X(1, "v_xor_b32 %0, s4, %0") X(2, "v_and_b32 %0, %1, %0") X(3, "v_and_b32 %0, s4, %0") \
	X(4, "v_or_b32 %0, %1, %0") X(5, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0xc8") X(6, "v_bitop3_b32 %0, %1, s4, %0 bitop3:0xc8") \
	X(7, "v_bcnt_u32_b32 %0, %1, %0") X(8, "v_bcnt_u32_b32 %0, s4, %0") X(9, "v_and_or_b32 %0, %1, %2, %0") X(10, "v_or3_b32 %0, %1, %2, %0") \
	X(11, "v_mov_b32 %0, %1") X(12, "v_mov_b32 %0, s4") X(13, "v_cndmask_b32 %0, %1, %0, vcc") X(14, "v_add_u32 %0, %1, %0") X(15, "v_sub_u32 %0, %1, %0") \
	X(16, "v_add3_u32 %0, %1, %2, %0") X(17, "v_lshlrev_b32 %0, 1, %0") X(18, "v_lshl_add_u32 %0, %1, 1, %0") X(19, "v_lshl_or_b32 %0, %1, 1, %0") \
	X(20, "v_min_u32 %0, %1, %0") X(21, "v_max_i32 %0, %1, %0") X(22, "v_mul_u32_u24 %0, %1, %0") X(23, "v_mad_u32_u24 %0, %1, %1, %0") X(24, "v_mul_lo_u32 %0, %1, %0") \
	X(25, "v_bfe_u32 %0, %0, 3, 8") X(26, "v_bfi_b32 %0, %1, %2, %0") X(27, "v_perm_b32 %0, %1, %0, %2") X(28, "v_alignbit_b32 %0, %1, %0, 5") \
	X(29, "v_pk_add_u16 %0, %1, %0") X(30, "v_cvt_f32_u32 %0, %0") X(31, "v_cvt_f32_ubyte1 %0, %0") X(32, "v_rcp_f32 %0, %0") X(33, "v_exp_f32 %0, %0") \
	X(34, "v_add_f32 %0, %1, %0") X(35, "v_mul_f32 %0, %1, %0") X(36, "v_fma_f32 %0, %1, %2, %0") X(37, "v_min_f32 %0, %1, %0") \
	X(39, "v_cmp_lt_u32 vcc, %1, %0") X(40, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") X(41, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
	X(42, "v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") X(43, "v_sad_u32 %0, %1, %2, %0") X(44, "v_mbcnt_lo_u32_b32 %0, %1, %0") X(45, "v_readlane_b32 s6, %0, 3") X(46, "v_cndmask_b32_e64 %0, %1, %0, s[8:9]") X(47, "v_not_b32 %0, %0") X(48, "v_xor_b32 %0, 0x12345678, %0") X(49, "v_xor_b32 %0, 5, %0") X(50, "v_fmac_f32 %0, %1, %2") X(51, "v_max_f32 %0, %1, %0") X(52, "v_mul_f32 %0, 0x40490fdb, %0") X(53, "v_mul_f32 %0, s4, %0")
#define OPS64(X) \
	X(100, "v_add_f64 %0, %1, %0") X(101, "v_mul_f64 %0, %1, %0") X(102, "v_fma_f64 %0, %1, %2, %0") X(103, "v_min_f64 %0, %1, %0") X(104, "v_rcp_f64 %0, %0") \
	X(105, "v_lshlrev_b64 %0, 1, %0") X(107, "v_cmp_lt_f64 vcc, %1, %0") X(109, "v_ldexp_f64 %0, %0, 1") X(110, "v_max_f64 %0, %1, %0") X(111, "v_fma_f64 %0, %1, s[8:9], %0") X(113, "v_mul_f64 %0, 2.0, %0")

static int g_clk = 0;
template<int OP>
__global__ __launch_bounds__(256) void k32(uint32_t* out, int iters, uint32_t seed) {
	uint32_t a[16];
	for(int i = 0; i < 16; ++i) a[i] = seed * (i + 1) + threadIdx.x;
	uint32_t x = seed ^ threadIdx.x, y = seed * 3 + blockIdx.x;
	asm volatile("s_mov_b32 s4, %0" : : "s"(seed) : "s4");
	for(int it = 0; it < iters; ++it) {
#define X(id, txt) if(OP == id) { _Pragma("unroll") for(int h = 0; h < 2; ++h) { _Pragma("unroll") for(int i = 0; i < 16; ++i) asm volatile(txt : "+v"(a[i]) : "v"(x), "v"(y) : "s4", "s6", "s8", "s9", "vcc"); } }
		OPS32(X)
#undef X
	}
	uint32_t s = y;
	for(int i = 0; i < 16; ++i) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template<int OP>
__global__ __launch_bounds__(256) void k64(uint32_t* out, int iters, uint32_t seed) {
	double a[16];
	for(int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-9 * (seed * (i + 1) + threadIdx.x);
	double x = 1.0 + 1e-12 * (seed ^ threadIdx.x), y = 1e-13 * blockIdx.x;
	for(int it = 0; it < iters; ++it) {
#define X(id, txt) if(OP == id) { _Pragma("unroll") for(int h = 0; h < 2; ++h) { _Pragma("unroll") for(int i = 0; i < 16; ++i) asm volatile(txt : "+v"(a[i]) : "v"(x), "v"(y) : "vcc", "s8", "s9"); } }
		OPS64(X)
#undef X
	}
	double s = y;
	for(int i = 0; i < 16; ++i) s += a[i];
	out[blockIdx.x * 256 + threadIdx.x] = (uint32_t) __double2loint(s);
}


template<class K>
static void run(K kern, const char* name, int wavesPerSimd) {
	const int iters = 2000, perIter = 32;
	const int blocks = 256 * wavesPerSimd;                 /* 4 waves per block: one per SIMD; `wavesPerSimd` blocks per CU */
	uint32_t* out; CHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	kern<<<blocks, 256>>>(out, 10, 1u);
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for(int r = 0; r < 3; ++r) {
		CHK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters, 12345u + r); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
	}
	const double instrPerSimd = (double) iters * perIter * wavesPerSimd;
	printf("%-72s %d waves/SIMD  %6.2f cycles per wave-instruction\n", name, wavesPerSimd, best * 1e-3 * g_clk * 1e3 / instrPerSimd);
	CHK(hipFree(out)); CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}


/* the (node, read, 32 sites) step of the seed scans as instruction mixes: cycles per STEP */
template<int MIX>
__global__ __launch_bounds__(256) void kmix(uint32_t* out, int iters, uint32_t seed) {
	uint32_t a[16], b[16];
	for(int i = 0; i < 16; ++i) { a[i] = seed * (i + 1) + threadIdx.x; b[i] = a[i] ^ 77u; }
	uint32_t x = seed ^ threadIdx.x, y = seed * 3 + blockIdx.x, z = seed * 7 + threadIdx.x * 3;
	asm volatile("s_mov_b32 s4, %0\n s_add_u32 s5, %0, 17\n s_add_u32 s6, %0, 99" : : "s"(seed) : "s4", "s5", "s6");
	for(int it = 0; it < iters; ++it) {
#pragma unroll
		for(int i = 0; i < 16; ++i) {
			uint32_t t1, t2, t3;
			if(MIX == 0) asm volatile("v_and_b32 %2, s4, %5\n v_xor_b32 %3, s5, %6\n v_xor_b32 %4, s6, %7\n v_bitop3_b32 %3, %4, %2, %3 bitop3:0xc8\n v_bcnt_u32_b32 %0, %3, %0\n v_bcnt_u32_b32 %1, %2, %1"
				: "+v"(a[i]), "+v"(b[i]), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x), "v"(y), "v"(z) : "s4", "s5", "s6");
			if(MIX == 1) asm volatile("v_xor_b32 %2, s4, %5\n v_bitop3_b32 %2, %2, %6, s5 bitop3:0xf6\n v_bitop3_b32 %2, %2, %7, s6 bitop3:0x80\n v_bcnt_u32_b32 %0, %2, %0"
				: "+v"(a[i]), "+v"(b[i]), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x), "v"(y), "v"(z) : "s4", "s5", "s6");
			if(MIX == 2) asm volatile("v_xor_b32 %2, %5, %6\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0xf6\n v_bitop3_b32 %2, %2, %7, %5 bitop3:0x80\n v_bcnt_u32_b32 %0, %2, %0"
				: "+v"(a[i]), "+v"(b[i]), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x), "v"(y), "v"(z) : "s4", "s5", "s6");
			if(MIX == 3) asm volatile("v_xor_b32 %2, %5, %6\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0xf6\n v_bitop3_b32 %2, %2, %7, %5 bitop3:0x80\n v_add_u32 %0, %2, %0"
				: "+v"(a[i]), "+v"(b[i]), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x), "v"(y), "v"(z) : "s4", "s5", "s6");
		}
	}
	uint32_t s2 = y;
	for(int i = 0; i < 16; ++i) s2 ^= a[i] ^ b[i];
	out[blockIdx.x * 256 + threadIdx.x] = s2;
}
template<class K>
static void runmix(K kern, const char* name, int wavesPerSimd) {
	const int iters = 2000, perIter = 16;
	const int blocks = 256 * wavesPerSimd;
	uint32_t* out; CHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	kern<<<blocks, 256>>>(out, 10, 1u);
	CHK(hipDeviceSynchronize());
	float best = 1e30f;
	for(int r = 0; r < 3; ++r) {
		CHK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters, 12345u + r); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if(ms < best) best = ms;
	}
	printf("%-72s %d waves/SIMD  %6.2f cycles per step\n", name, wavesPerSimd, best * 1e-3 * g_clk * 1e3 / ((double) iters * perIter * wavesPerSimd));
	CHK(hipFree(out)); CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main() {
	CHK(hipDeviceGetAttribute(&g_clk, hipDeviceAttributeClockRate, 0));
	printf("nominal clock %d MHz; cycles below are at that clock (the chip may run lower under load)\n", g_clk / 1000);
	for(int w : {8, 4}) {
		runmix(kmix<0>, "scan step with N: and(s) xor(s) xor(s) bitop3 bcnt bcnt", w);
		runmix(kmix<1>, "scan step, d only: xor(s) bitop3(s) bitop3(s) bcnt", w);
		runmix(kmix<2>, "the same on vector registers only", w);
		runmix(kmix<3>, "the same with v_add_u32 in place of v_bcnt", w);
	}
	for(int w : {8, 2, 1}) {
#define X(id, txt) run(k32<id>, txt, w);
		OPS32(X)
#undef X
#define X(id, txt) run(k64<id>, txt, w);
		OPS64(X)
#undef X
	}
	return 0;
}
