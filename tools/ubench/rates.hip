// VALU instruction issue rates on gfx950 (cycles per wave64 instruction per SIMD, saturated)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short ss2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int n, int a0, int b0) {
    int r0 = threadIdx.x + a0, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 + 11, r5 = r0 + 13, r6 = r0 + 17, r7 = r0 + 19;
    const int a = a0 | 0x10001, b = b0 | 0x20003;
    float f0 = r0, f1 = r1, f2 = r2, f3 = r3, f4 = r4, f5 = r5, f6 = r6, f7 = r7;
    const float fa = a0 * 1e-3f + 1.0001f, fb = b0 * 1e-3f + 0.5f;
    double d0 = r0, d1 = r1, d2 = r2, d3 = r3;
    const double da = a0 * 1e-3 + 1.0001, db = b0 * 1e-3 + 0.5;
    for (int i = 0; i < n; ++i) {
        if (OP == 0) { REP8(r0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), r0, false); r1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), r1, false); r2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), r2, false); r3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2, a), __builtin_bit_cast(ss2, b), r3, false);) }
        if (OP == 1) { REP8(r0 = __mul24(r0, a) + b; r1 = __mul24(r1, a) + b; r2 = __mul24(r2, a) + b; r3 = __mul24(r3, a) + b;) }
        if (OP == 2) { REP8(r0 = r0 * a + b; r1 = r1 * a + b; r2 = r2 * a + b; r3 = r3 * a + b;) }
        if (OP == 3) { REP8(f0 = f0 * fa + fb; f1 = f1 * fa + fb; f2 = f2 * fa + fb; f3 = f3 * fa + fb;) }
        if (OP == 4) { REP8(r0 = (r0 >> 9) - a; r1 = (r1 >> 9) - a; r2 = (r2 >> 9) - a; r3 = (r3 >> 9) - a;) }
        if (OP == 5) { REP8(r0 = __builtin_amdgcn_perm(r0, a, 0x05040100u) ^ b; r1 = __builtin_amdgcn_perm(r1, a, 0x05040100u) ^ b; r2 = __builtin_amdgcn_perm(r2, a, 0x05040100u) ^ b; r3 = __builtin_amdgcn_perm(r3, a, 0x05040100u) ^ b;) }
        if (OP == 6) { REP8(f0 = __builtin_rintf(f0 * fa); f1 = __builtin_rintf(f1 * fa); f2 = __builtin_rintf(f2 * fa); f3 = __builtin_rintf(f3 * fa);) }
        if (OP == 7) { REP8(r0 += __builtin_amdgcn_update_dpp(0, r0, 0x128, 0xF, 0xF, false); r1 += __builtin_amdgcn_update_dpp(0, r1, 0x124, 0xF, 0xF, false); r2 += __builtin_amdgcn_update_dpp(0, r2, 0x122, 0xF, 0xF, false); r3 += __builtin_amdgcn_update_dpp(0, r3, 0x121, 0xF, 0xF, false);) }
        if (OP == 8) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b));) }
        if (OP == 9) { REP8(asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r0) : "v"(a)); asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r1) : "v"(a)); asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r2) : "v"(a)); asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(r3) : "v"(a));) }
        if (OP == 10) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b));) }
        if (OP == 11) { REP8(asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r0) : "v"(a)); asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r1) : "v"(a)); asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r2) : "v"(a)); asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(r3) : "v"(a));) }
        if (OP == 12) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r0) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r1) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r2) : "v"(a)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r3) : "v"(a));) }
        if (OP == 13) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r0) : "v"(a)); asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r1) : "v"(a)); asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r2) : "v"(a)); asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r3) : "v"(a));) }
        if (OP == 14) { REP8(asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r0)); asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r1)); asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r2)); asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r3));) }
        if (OP == 15) { REP8(asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r0)); asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r1)); asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r2)); asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(r3));) }
        if (OP == 16) { REP8(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r0) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r1) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r2) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r3) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r4) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r5) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r6) : "v"(a)); asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r7) : "v"(a));) }
        if (OP == 17) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r4) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r5) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r6) : "v"(a), "v"(b)); asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(r7) : "v"(a), "v"(b));) }
        if (OP == 18) { REP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d1) : "v"(da)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d2) : "v"(da)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d3) : "v"(da));) }
        if (OP == 19) { REP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(da), "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(da), "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(da), "v"(db)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(da), "v"(db));) }
        if (OP == 20) { REP8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(f0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(f1)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(f2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(f3));) }
        if (OP == 21) { REP8(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(d0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f1) : "v"(d1)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f2) : "v"(d2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f3) : "v"(d3));) }
        if (OP == 22) { REP8(asm volatile("v_sqrt_f32 %0, %0" : "+v"(f0)); asm volatile("v_sqrt_f32 %0, %0" : "+v"(f1)); asm volatile("v_sqrt_f32 %0, %0" : "+v"(f2)); asm volatile("v_sqrt_f32 %0, %0" : "+v"(f3));) }
        if (OP == 23) { REP8(asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f0) : "v"(r0)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f1) : "v"(r1)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f2) : "v"(r2)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f3) : "v"(r3));) }
        if (OP == 24) { REP8(asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r3) : "v"(a), "v"(b));) }
        if (OP == 25) { REP8(asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r0)); asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r1)); asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r2)); asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r3));) }
        if (OP == 26) { REP8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(fa)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f1) : "v"(fa)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f2) : "v"(fa)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f3) : "v"(fa));) }
        if (OP == 27) { REP8(asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(r0) : "v"(a)); asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(r1) : "v"(a)); asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(r2) : "v"(a)); asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(r3) : "v"(a));) }
        if (OP == 28) { REP8(asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(r3) : "v"(a), "v"(b));) }
        if (OP == 29) { REP8(asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r0) : "v"(a), "v"(b), "v"(r4)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r1) : "v"(a), "v"(b), "v"(r5)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r2) : "v"(a), "v"(b), "v"(r6)); asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r3) : "v"(a), "v"(b), "v"(r7));) }
        if (OP == 31) { REP8(asm volatile("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r0) : "v"(a), "v"(r4)); asm volatile("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r1) : "v"(a), "v"(r5)); asm volatile("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r2) : "v"(a), "v"(r6)); asm volatile("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r3) : "v"(a), "v"(r7));) }
        if (OP == 32) { REP8(asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(r0) : "v"(r4)); asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(r1) : "v"(r5)); asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(r2) : "v"(r6)); asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(r3) : "v"(r7));) }
        if (OP == 33) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b));) }
        if (OP == 30) { REP8(asm volatile("v_mov_b32 %0, %1" : "=v"(r0) : "v"(r4)); asm volatile("v_mov_b32 %0, %1" : "=v"(r1) : "v"(r5)); asm volatile("v_mov_b32 %0, %1" : "=v"(r2) : "v"(r6)); asm volatile("v_mov_b32 %0, %1" : "=v"(r3) : "v"(r7));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (int)(d0 + d1 + d2 + d3) + r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (int)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}
template <int OP> void run(const char* name, int* d, int ops_per_iter) {
    const int blocks = 256 * 8, n = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n, 1, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n, 1, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = blocks*4/1024 ; instr per wave = n*ops_per_iter
    const double instr_per_simd = (double)blocks * 4 / 1024 * n * ops_per_iter;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / instr_per_simd);
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("v_fma_f32 (reference)", d, 32);
    run<0>("v_dot2c_i32_i16", d, 32);
    run<1>("v_mul_i32_i24 + v_add (2 ops)", d, 64);
    run<2>("v_mul_lo_u32 + add (2 ops)", d, 64);
    run<4>("v_ashr + v_sub (2 ops)", d, 64);
    run<5>("v_perm + v_xor (2 ops)", d, 64);
    run<6>("v_mul_f32 + v_rndne (2 ops)", d, 64);
    run<7>("v_add_dpp (+nops)", d, 32);
    run<8>("v_mad_i32_i24", d, 32);
    run<9>("v_mul_i32_i24", d, 32);
    run<10>("v_add3_u32", d, 32);
    run<11>("v_pack_b32_f16", d, 32);
    run<12>("v_alignbit_b32", d, 32);
    run<13>("v_lshl_or_b32", d, 32);
    run<14>("v_cvt_f32_ubyte1", d, 32);
    run<15>("asm v_ashrrev (4 chains)", d, 32);
    run<16>("asm v_sub_u32 (8 chains)", d, 64);
    run<17>("asm v_mad_i32_i24 (8 chains)", d, 64);
    run<18>("v_add_f64", d, 32);
    run<19>("v_fma_f64", d, 32);
    run<20>("v_cvt_f64_f32", d, 32);
    run<21>("v_cvt_f32_f64", d, 32);
    run<22>("v_sqrt_f32", d, 32);
    run<23>("v_cvt_f32_i32", d, 32);
    run<24>("v_dot4_u32_u8", d, 32);
    run<25>("v_bfe_u32", d, 32);
    run<26>("v_mul_f32", d, 32);
    run<27>("v_alignbyte_b32", d, 32);
    run<28>("v_dot2_i32_i16 VOP3P acc", d, 32);
    run<29>("v_dot2_i32_i16 VOP3P seed", d, 32);
    run<30>("v_mov_b32", d, 32);
    run<31>("v_ashrrev_i32_sdwa (into WORD_1, preserve)", d, 32);
    run<32>("v_mov_b32_sdwa (WORD_0 -> WORD_1, preserve)", d, 32);
    run<33>("v_perm_b32 (4 chains)", d, 32);
    return 0;
}
