// valu_rates.hip -- measurement tool (not part of the product library): issue cost of the VALU instruction classes the
// engine's hot loops are made of, measured on the box the benchmark runs on.
//
// `roofline.valu_frac` priced every vector instruction at a flat 4 cycles "on a SIMD16".  The guide says
// a CU has 4 SIMD-32 units and a wave64 32-bit VALU op issues over 2 cycles (MI355X_MICROARCH.md "Wave scheduling"); binary64
// arithmetic runs at half that rate, v_mad_u64_u32 / v_mul_hi_u32 at a quarter.  Instead of trusting a table this tool MEASURES each
// class: a kernel of straight-line inline assembly, independent dependency chains, enough wavefronts per SIMD to hide every latency,
// so that the time is pure issue.  Output: ns x SIMD per wave-instruction (wall clock, hipEvents) and the same in cycles of the
// shader clock (s_memtime ticks of one wavefront divided by the instructions every wavefront of its SIMD issued in that span).
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates > profiles/r06_valu_rates.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kUnroll = 64;     // instructions per chain group and loop trip (8 chains x 8)

// one loop trip = kUnroll instructions of the class: 8 independent accumulators, 8 instructions each
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define TRIP(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

enum Cls {
    C_ADD_U32, C_XOR_B32, C_MUL_LO_U32, C_MUL_HI_U32, C_MAD_U64_U32, C_ADD_F64, C_MUL_F64, C_FMA_F64, C_MAX_F64, C_TRUNC_F64, C_RCP_F64,
    C_LDEXP_F64, C_DIV_SCALE_F64, C_DIV_FMAS_F64, C_DIV_FIXUP_F64, C_CVT_F64_I32, C_CVT_I32_F64, C_CMP_F64, C_CMP_I64, C_CMP_U32, C_CNDMASK,
    C_ADD_CO_PAIR, C_LSHL_ADD_U64, C_MOV_B32, C_MOV_DPP, C_LSHLREV_B64, C_ALIGNBIT, C_BFE, C_CMP_CNDMASK, C_N
};
static const char *kNames[C_N] = {
    "v_add_u32", "v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_trunc_f64",
    "v_rcp_f64", "v_ldexp_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_cmp_lt_f64",
    "v_cmp_lt_i64", "v_cmp_lt_u32", "v_cndmask_b32", "v_add_co_u32+v_addc_co_u32 (per instruction)", "v_lshl_add_u64", "v_mov_b32",
    "v_mov_b32_dpp row_shr:1", "v_lshlrev_b64", "v_alignbit_b32", "v_bfe_u32", "v_cmp_lt_u32+v_cndmask_b32 (dependent pair, per instruction)"};

template <int CLS>
__global__ void __launch_bounds__(256) rate_kernel(unsigned long long *out, int trips, double seed_d, unsigned seed_u) {
    double d[8];
    unsigned u[8];
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = seed_d + i + threadIdx.x * 1e-3; u[i] = seed_u + i * 7u + threadIdx.x; q[i] = ((unsigned long long)u[i] << 20) + i; }
    double dk = 1.0000001; unsigned uk = 0x9e3779b9u; unsigned long long qk = 0x12345ull;
    unsigned long long mask = 0x5555aaaa3333ccccull, sel[4] = {0, 0, 0, 0};
    asm volatile("" : "+s"(mask));
    asm volatile("" : "+v"(dk), "+v"(uk), "+v"(qk));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
        if constexpr (CLS == C_ADD_U32) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_XOR_B32) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MUL_LO_U32) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MUL_HI_U32) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MAD_U64_U32) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(u[i]), "v"(uk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_ADD_F64) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MUL_F64) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_FMA_F64) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MAX_F64) {
#define S(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_TRUNC_F64) {
#define S(i) asm volatile("v_trunc_f64 %0, %0" : "+v"(d[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_RCP_F64) {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_LDEXP_F64) {
#define S(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i] & 1u));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_DIV_SCALE_F64) {
#define S(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[i]) : "v"(dk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_DIV_FMAS_F64) {
#define S(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_DIV_FIXUP_F64) {
#define S(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CVT_F64_I32) {
#define S(i) asm volatile("v_cvt_f64_i32 %0, %1" : "+v"(d[i]) : "v"(u[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CVT_I32_F64) {
#define S(i) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(u[i]) : "v"(d[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CMP_F64) {
#define S(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(dk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CMP_I64) {
#define S(i) asm volatile("v_cmp_lt_i64 vcc, %0, %1" : : "v"(q[i]), "v"(qk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CMP_U32) {
#define S(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(uk) : "vcc");
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CNDMASK) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(uk), "s"(mask));   // (the condition in an SGPR pair nobody writes)
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_CMP_CNDMASK) {   // a select as the compiler writes it: v_cmp into an SGPR pair, v_cndmask reads it (32 pairs per trip)
#define S(i) asm volatile("v_cmp_lt_u32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(u[i]), "=&s"(sel[i & 3]) : "v"(uk));
            REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
        } else if constexpr (CLS == C_ADD_CO_PAIR) {    // a 64-bit add as the compiler writes it: two instructions (32 pairs per trip)
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(u[i]), "+v"(u[(i + 4) & 7]) : "v"(uk) : "vcc");
            REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
        } else if constexpr (CLS == C_LSHL_ADD_U64) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[i]) : "v"(qk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MOV_B32) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_MOV_DPP) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_LSHLREV_B64) {
#define S(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_ALIGNBIT) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(u[i]) : "v"(uk));
            TRIP(S)
#undef S
        } else if constexpr (CLS == C_BFE) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(u[i]));
            TRIP(S)
#undef S
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += (unsigned long long)u[i] + q[i] + (unsigned long long)__double_as_longlong(d[i]);
    acc += sel[0] + sel[1] + sel[2] + sel[3];
    if (acc == 0x1234567887654321ull) out[1] = acc;                     // (keeps the chains alive)
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;           // shader-clock ticks of one wavefront
}

template <int CLS>
static void run_one(int cls, int blocks_per_cu, int n_cu, unsigned long long *d_out, FILE *f, bool first) {
    const int trips = 4096;
    const int blocks = blocks_per_cu * n_cu;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(256), 0, 0, d_out, 64, 1.5, 12345u);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(rate_kernel<CLS>, dim3(blocks), dim3(256), 0, 0, d_out, trips, 1.5, 12345u);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long ticks = 0;
    CK(hipMemcpy(&ticks, d_out, sizeof ticks, hipMemcpyDeviceToHost));
    const double per_wave = (double)trips * kUnroll;                     // instructions one wavefront issued
    const double waves_per_simd = blocks_per_cu;                         // a 256-thread block puts one wavefront on each of the CU's 4 SIMDs
    const double ns_simd = (double)ms * 1e6 / (per_wave * waves_per_simd);          // ns of one SIMD per wave-instruction
    const double cyc = (double)ticks / (per_wave * waves_per_simd);                 // shader-clock ticks of one SIMD per wave-instruction
    fprintf(f, "%s\n    {\"class\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_inst_per_simd\": %.4f, \"memtime_ticks_per_wave_inst\": %.3f, \"kernel_ms\": %.4f}",
            first ? "" : ",", kNames[cls], blocks_per_cu, ns_simd, cyc, ms);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

template <int CLS>
static void run_all(int n_cu, unsigned long long *d_out, FILE *f, bool &first) {
    if constexpr (CLS < C_N) {
        for (int w : {8, 2, 1}) { run_one<CLS>(CLS, w, n_cu, d_out, f, first); first = false; }
        run_all<CLS + 1>(n_cu, d_out, f, first);
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, 64));
    CK(hipMemset(d_out, 0, 64));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d,\n \"note\": \"straight-line inline assembly, 8 independent chains per wavefront, "
           "W wavefronts per SIMD on every SIMD of the device; ns_per_wave_inst_per_simd = kernel time / (instructions per wavefront x W); "
           "memtime ticks: s_memtime of one wavefront over the same span (100 MHz-class constant clock on gfx9: a cross-check of the wall clock only)\",\n \"rates\": [",
           prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    bool first = true;
    run_all<0>(prop.multiProcessorCount, d_out, stdout, first);
    printf("\n]}\n");
    return 0;
}
