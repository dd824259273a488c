// r04: synthetic "neighbours" for the decoder-core hunt: each kernel stresses ONE resource a coder wave shares
// with the transforms -- the matrix pipe, the vector ALU, LDS, the memory system -- so that the failing decoder form can be run next
// to each of them in turn.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/liblk.so scratch/r04/load_kernels.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float16_t __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void lk_mfma(float* sink, int iters, float seed) {
    float16_t acc0 = {0}, acc1 = {0};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        a += 1e-7f; b -= 1e-7f;
    }
    float s = 0;
    for (int k = 0; k < 16; k++) s += acc0[k] + acc1[k];
    if (s == 12345.678f) sink[0] = s;
}
// many registers, like the GEMM waves (forces the same occupancy pattern): VALU fma chains
__global__ __launch_bounds__(256) void lk_valu(float* sink, int iters, float seed) {
    float v[32];
    for (int k = 0; k < 32; k++) v[k] = seed + k + threadIdx.x;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = __builtin_fmaf(v[k], 1.0000001f, 1e-9f);
    float s = 0;
    for (int k = 0; k < 32; k++) s += v[k];
    if (s == 12345.678f) sink[0] = s;
}
// LDS: every wave hammers its own slice with the GEMM's access kinds (b128 reads, b64 writes); static size like the GEMM's block
__global__ __launch_bounds__(256) void lk_lds(float* sink, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[9728];           // 38 KB, the conv GEMM's block
    float* mine = lds + (threadIdx.x >> 6) * 2432;
    const int lane = threadIdx.x & 63;
    for (int k = lane; k < 2432; k += 64) mine[k] = seed + k;
    float4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 r = *reinterpret_cast<const float4*>(mine + ((lane * 4 + k * 256 + i * 4) % 2400 & ~3));
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        *reinterpret_cast<float2*>(mine + ((lane * 2 + i * 2) % 2400 & ~1)) = make_float2(acc.x, acc.y);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
// memory: a streaming copy with 16-byte accesses
__global__ __launch_bounds__(256) void lk_mem(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// LDS written WITHOUT a matching allocation check: fills the block's whole static allocation with a pattern and leaves it there
// (what does a coder block that starts on this CU afterwards find in its ring / probability rows?)
__global__ __launch_bounds__(256) void lk_lds_dirty(float* sink, unsigned pattern) {
    __shared__ unsigned lds[16384];                                    // 64 KB
    for (int k = threadIdx.x; k < 16384; k += 256) lds[k] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 7) & 16383] != pattern) sink[1] = 1.f;
}

// every vector register of the SIMD left holding a NaN pattern (256 arch VGPRs per lane; the next wave in that slot finds them)
__global__ __launch_bounds__(64) void lk_vgpr_dirty(float* sink, float seed) {
    asm volatile("v_mov_b32 v8, 0x7fc0beef" ::: "v8");
    asm volatile("v_mov_b32 v9, 0x7fc0beef" ::: "v9");
    asm volatile("v_mov_b32 v10, 0x7fc0beef" ::: "v10");
    asm volatile("v_mov_b32 v11, 0x7fc0beef" ::: "v11");
    asm volatile("v_mov_b32 v12, 0x7fc0beef" ::: "v12");
    asm volatile("v_mov_b32 v13, 0x7fc0beef" ::: "v13");
    asm volatile("v_mov_b32 v14, 0x7fc0beef" ::: "v14");
    asm volatile("v_mov_b32 v15, 0x7fc0beef" ::: "v15");
    asm volatile("v_mov_b32 v16, 0x7fc0beef" ::: "v16");
    asm volatile("v_mov_b32 v17, 0x7fc0beef" ::: "v17");
    asm volatile("v_mov_b32 v18, 0x7fc0beef" ::: "v18");
    asm volatile("v_mov_b32 v19, 0x7fc0beef" ::: "v19");
    asm volatile("v_mov_b32 v20, 0x7fc0beef" ::: "v20");
    asm volatile("v_mov_b32 v21, 0x7fc0beef" ::: "v21");
    asm volatile("v_mov_b32 v22, 0x7fc0beef" ::: "v22");
    asm volatile("v_mov_b32 v23, 0x7fc0beef" ::: "v23");
    asm volatile("v_mov_b32 v24, 0x7fc0beef" ::: "v24");
    asm volatile("v_mov_b32 v25, 0x7fc0beef" ::: "v25");
    asm volatile("v_mov_b32 v26, 0x7fc0beef" ::: "v26");
    asm volatile("v_mov_b32 v27, 0x7fc0beef" ::: "v27");
    asm volatile("v_mov_b32 v28, 0x7fc0beef" ::: "v28");
    asm volatile("v_mov_b32 v29, 0x7fc0beef" ::: "v29");
    asm volatile("v_mov_b32 v30, 0x7fc0beef" ::: "v30");
    asm volatile("v_mov_b32 v31, 0x7fc0beef" ::: "v31");
    asm volatile("v_mov_b32 v32, 0x7fc0beef" ::: "v32");
    asm volatile("v_mov_b32 v33, 0x7fc0beef" ::: "v33");
    asm volatile("v_mov_b32 v34, 0x7fc0beef" ::: "v34");
    asm volatile("v_mov_b32 v35, 0x7fc0beef" ::: "v35");
    asm volatile("v_mov_b32 v36, 0x7fc0beef" ::: "v36");
    asm volatile("v_mov_b32 v37, 0x7fc0beef" ::: "v37");
    asm volatile("v_mov_b32 v38, 0x7fc0beef" ::: "v38");
    asm volatile("v_mov_b32 v39, 0x7fc0beef" ::: "v39");
    asm volatile("v_mov_b32 v40, 0x7fc0beef" ::: "v40");
    asm volatile("v_mov_b32 v41, 0x7fc0beef" ::: "v41");
    asm volatile("v_mov_b32 v42, 0x7fc0beef" ::: "v42");
    asm volatile("v_mov_b32 v43, 0x7fc0beef" ::: "v43");
    asm volatile("v_mov_b32 v44, 0x7fc0beef" ::: "v44");
    asm volatile("v_mov_b32 v45, 0x7fc0beef" ::: "v45");
    asm volatile("v_mov_b32 v46, 0x7fc0beef" ::: "v46");
    asm volatile("v_mov_b32 v47, 0x7fc0beef" ::: "v47");
    asm volatile("v_mov_b32 v48, 0x7fc0beef" ::: "v48");
    asm volatile("v_mov_b32 v49, 0x7fc0beef" ::: "v49");
    asm volatile("v_mov_b32 v50, 0x7fc0beef" ::: "v50");
    asm volatile("v_mov_b32 v51, 0x7fc0beef" ::: "v51");
    asm volatile("v_mov_b32 v52, 0x7fc0beef" ::: "v52");
    asm volatile("v_mov_b32 v53, 0x7fc0beef" ::: "v53");
    asm volatile("v_mov_b32 v54, 0x7fc0beef" ::: "v54");
    asm volatile("v_mov_b32 v55, 0x7fc0beef" ::: "v55");
    asm volatile("v_mov_b32 v56, 0x7fc0beef" ::: "v56");
    asm volatile("v_mov_b32 v57, 0x7fc0beef" ::: "v57");
    asm volatile("v_mov_b32 v58, 0x7fc0beef" ::: "v58");
    asm volatile("v_mov_b32 v59, 0x7fc0beef" ::: "v59");
    asm volatile("v_mov_b32 v60, 0x7fc0beef" ::: "v60");
    asm volatile("v_mov_b32 v61, 0x7fc0beef" ::: "v61");
    asm volatile("v_mov_b32 v62, 0x7fc0beef" ::: "v62");
    asm volatile("v_mov_b32 v63, 0x7fc0beef" ::: "v63");
    asm volatile("v_mov_b32 v64, 0x7fc0beef" ::: "v64");
    asm volatile("v_mov_b32 v65, 0x7fc0beef" ::: "v65");
    asm volatile("v_mov_b32 v66, 0x7fc0beef" ::: "v66");
    asm volatile("v_mov_b32 v67, 0x7fc0beef" ::: "v67");
    asm volatile("v_mov_b32 v68, 0x7fc0beef" ::: "v68");
    asm volatile("v_mov_b32 v69, 0x7fc0beef" ::: "v69");
    asm volatile("v_mov_b32 v70, 0x7fc0beef" ::: "v70");
    asm volatile("v_mov_b32 v71, 0x7fc0beef" ::: "v71");
    asm volatile("v_mov_b32 v72, 0x7fc0beef" ::: "v72");
    asm volatile("v_mov_b32 v73, 0x7fc0beef" ::: "v73");
    asm volatile("v_mov_b32 v74, 0x7fc0beef" ::: "v74");
    asm volatile("v_mov_b32 v75, 0x7fc0beef" ::: "v75");
    asm volatile("v_mov_b32 v76, 0x7fc0beef" ::: "v76");
    asm volatile("v_mov_b32 v77, 0x7fc0beef" ::: "v77");
    asm volatile("v_mov_b32 v78, 0x7fc0beef" ::: "v78");
    asm volatile("v_mov_b32 v79, 0x7fc0beef" ::: "v79");
    asm volatile("v_mov_b32 v80, 0x7fc0beef" ::: "v80");
    asm volatile("v_mov_b32 v81, 0x7fc0beef" ::: "v81");
    asm volatile("v_mov_b32 v82, 0x7fc0beef" ::: "v82");
    asm volatile("v_mov_b32 v83, 0x7fc0beef" ::: "v83");
    asm volatile("v_mov_b32 v84, 0x7fc0beef" ::: "v84");
    asm volatile("v_mov_b32 v85, 0x7fc0beef" ::: "v85");
    asm volatile("v_mov_b32 v86, 0x7fc0beef" ::: "v86");
    asm volatile("v_mov_b32 v87, 0x7fc0beef" ::: "v87");
    asm volatile("v_mov_b32 v88, 0x7fc0beef" ::: "v88");
    asm volatile("v_mov_b32 v89, 0x7fc0beef" ::: "v89");
    asm volatile("v_mov_b32 v90, 0x7fc0beef" ::: "v90");
    asm volatile("v_mov_b32 v91, 0x7fc0beef" ::: "v91");
    asm volatile("v_mov_b32 v92, 0x7fc0beef" ::: "v92");
    asm volatile("v_mov_b32 v93, 0x7fc0beef" ::: "v93");
    asm volatile("v_mov_b32 v94, 0x7fc0beef" ::: "v94");
    asm volatile("v_mov_b32 v95, 0x7fc0beef" ::: "v95");
    asm volatile("v_mov_b32 v96, 0x7fc0beef" ::: "v96");
    asm volatile("v_mov_b32 v97, 0x7fc0beef" ::: "v97");
    asm volatile("v_mov_b32 v98, 0x7fc0beef" ::: "v98");
    asm volatile("v_mov_b32 v99, 0x7fc0beef" ::: "v99");
    asm volatile("v_mov_b32 v100, 0x7fc0beef" ::: "v100");
    asm volatile("v_mov_b32 v101, 0x7fc0beef" ::: "v101");
    asm volatile("v_mov_b32 v102, 0x7fc0beef" ::: "v102");
    asm volatile("v_mov_b32 v103, 0x7fc0beef" ::: "v103");
    asm volatile("v_mov_b32 v104, 0x7fc0beef" ::: "v104");
    asm volatile("v_mov_b32 v105, 0x7fc0beef" ::: "v105");
    asm volatile("v_mov_b32 v106, 0x7fc0beef" ::: "v106");
    asm volatile("v_mov_b32 v107, 0x7fc0beef" ::: "v107");
    asm volatile("v_mov_b32 v108, 0x7fc0beef" ::: "v108");
    asm volatile("v_mov_b32 v109, 0x7fc0beef" ::: "v109");
    asm volatile("v_mov_b32 v110, 0x7fc0beef" ::: "v110");
    asm volatile("v_mov_b32 v111, 0x7fc0beef" ::: "v111");
    asm volatile("v_mov_b32 v112, 0x7fc0beef" ::: "v112");
    asm volatile("v_mov_b32 v113, 0x7fc0beef" ::: "v113");
    asm volatile("v_mov_b32 v114, 0x7fc0beef" ::: "v114");
    asm volatile("v_mov_b32 v115, 0x7fc0beef" ::: "v115");
    asm volatile("v_mov_b32 v116, 0x7fc0beef" ::: "v116");
    asm volatile("v_mov_b32 v117, 0x7fc0beef" ::: "v117");
    asm volatile("v_mov_b32 v118, 0x7fc0beef" ::: "v118");
    asm volatile("v_mov_b32 v119, 0x7fc0beef" ::: "v119");
    asm volatile("v_mov_b32 v120, 0x7fc0beef" ::: "v120");
    asm volatile("v_mov_b32 v121, 0x7fc0beef" ::: "v121");
    asm volatile("v_mov_b32 v122, 0x7fc0beef" ::: "v122");
    asm volatile("v_mov_b32 v123, 0x7fc0beef" ::: "v123");
    asm volatile("v_mov_b32 v124, 0x7fc0beef" ::: "v124");
    asm volatile("v_mov_b32 v125, 0x7fc0beef" ::: "v125");
    asm volatile("v_mov_b32 v126, 0x7fc0beef" ::: "v126");
    asm volatile("v_mov_b32 v127, 0x7fc0beef" ::: "v127");
    asm volatile("v_mov_b32 v128, 0x7fc0beef" ::: "v128");
    asm volatile("v_mov_b32 v129, 0x7fc0beef" ::: "v129");
    asm volatile("v_mov_b32 v130, 0x7fc0beef" ::: "v130");
    asm volatile("v_mov_b32 v131, 0x7fc0beef" ::: "v131");
    asm volatile("v_mov_b32 v132, 0x7fc0beef" ::: "v132");
    asm volatile("v_mov_b32 v133, 0x7fc0beef" ::: "v133");
    asm volatile("v_mov_b32 v134, 0x7fc0beef" ::: "v134");
    asm volatile("v_mov_b32 v135, 0x7fc0beef" ::: "v135");
    asm volatile("v_mov_b32 v136, 0x7fc0beef" ::: "v136");
    asm volatile("v_mov_b32 v137, 0x7fc0beef" ::: "v137");
    asm volatile("v_mov_b32 v138, 0x7fc0beef" ::: "v138");
    asm volatile("v_mov_b32 v139, 0x7fc0beef" ::: "v139");
    asm volatile("v_mov_b32 v140, 0x7fc0beef" ::: "v140");
    asm volatile("v_mov_b32 v141, 0x7fc0beef" ::: "v141");
    asm volatile("v_mov_b32 v142, 0x7fc0beef" ::: "v142");
    asm volatile("v_mov_b32 v143, 0x7fc0beef" ::: "v143");
    asm volatile("v_mov_b32 v144, 0x7fc0beef" ::: "v144");
    asm volatile("v_mov_b32 v145, 0x7fc0beef" ::: "v145");
    asm volatile("v_mov_b32 v146, 0x7fc0beef" ::: "v146");
    asm volatile("v_mov_b32 v147, 0x7fc0beef" ::: "v147");
    asm volatile("v_mov_b32 v148, 0x7fc0beef" ::: "v148");
    asm volatile("v_mov_b32 v149, 0x7fc0beef" ::: "v149");
    asm volatile("v_mov_b32 v150, 0x7fc0beef" ::: "v150");
    asm volatile("v_mov_b32 v151, 0x7fc0beef" ::: "v151");
    asm volatile("v_mov_b32 v152, 0x7fc0beef" ::: "v152");
    asm volatile("v_mov_b32 v153, 0x7fc0beef" ::: "v153");
    asm volatile("v_mov_b32 v154, 0x7fc0beef" ::: "v154");
    asm volatile("v_mov_b32 v155, 0x7fc0beef" ::: "v155");
    asm volatile("v_mov_b32 v156, 0x7fc0beef" ::: "v156");
    asm volatile("v_mov_b32 v157, 0x7fc0beef" ::: "v157");
    asm volatile("v_mov_b32 v158, 0x7fc0beef" ::: "v158");
    asm volatile("v_mov_b32 v159, 0x7fc0beef" ::: "v159");
    asm volatile("v_mov_b32 v160, 0x7fc0beef" ::: "v160");
    asm volatile("v_mov_b32 v161, 0x7fc0beef" ::: "v161");
    asm volatile("v_mov_b32 v162, 0x7fc0beef" ::: "v162");
    asm volatile("v_mov_b32 v163, 0x7fc0beef" ::: "v163");
    asm volatile("v_mov_b32 v164, 0x7fc0beef" ::: "v164");
    asm volatile("v_mov_b32 v165, 0x7fc0beef" ::: "v165");
    asm volatile("v_mov_b32 v166, 0x7fc0beef" ::: "v166");
    asm volatile("v_mov_b32 v167, 0x7fc0beef" ::: "v167");
    asm volatile("v_mov_b32 v168, 0x7fc0beef" ::: "v168");
    asm volatile("v_mov_b32 v169, 0x7fc0beef" ::: "v169");
    asm volatile("v_mov_b32 v170, 0x7fc0beef" ::: "v170");
    asm volatile("v_mov_b32 v171, 0x7fc0beef" ::: "v171");
    asm volatile("v_mov_b32 v172, 0x7fc0beef" ::: "v172");
    asm volatile("v_mov_b32 v173, 0x7fc0beef" ::: "v173");
    asm volatile("v_mov_b32 v174, 0x7fc0beef" ::: "v174");
    asm volatile("v_mov_b32 v175, 0x7fc0beef" ::: "v175");
    asm volatile("v_mov_b32 v176, 0x7fc0beef" ::: "v176");
    asm volatile("v_mov_b32 v177, 0x7fc0beef" ::: "v177");
    asm volatile("v_mov_b32 v178, 0x7fc0beef" ::: "v178");
    asm volatile("v_mov_b32 v179, 0x7fc0beef" ::: "v179");
    asm volatile("v_mov_b32 v180, 0x7fc0beef" ::: "v180");
    asm volatile("v_mov_b32 v181, 0x7fc0beef" ::: "v181");
    asm volatile("v_mov_b32 v182, 0x7fc0beef" ::: "v182");
    asm volatile("v_mov_b32 v183, 0x7fc0beef" ::: "v183");
    asm volatile("v_mov_b32 v184, 0x7fc0beef" ::: "v184");
    asm volatile("v_mov_b32 v185, 0x7fc0beef" ::: "v185");
    asm volatile("v_mov_b32 v186, 0x7fc0beef" ::: "v186");
    asm volatile("v_mov_b32 v187, 0x7fc0beef" ::: "v187");
    asm volatile("v_mov_b32 v188, 0x7fc0beef" ::: "v188");
    asm volatile("v_mov_b32 v189, 0x7fc0beef" ::: "v189");
    asm volatile("v_mov_b32 v190, 0x7fc0beef" ::: "v190");
    asm volatile("v_mov_b32 v191, 0x7fc0beef" ::: "v191");
    asm volatile("v_mov_b32 v192, 0x7fc0beef" ::: "v192");
    asm volatile("v_mov_b32 v193, 0x7fc0beef" ::: "v193");
    asm volatile("v_mov_b32 v194, 0x7fc0beef" ::: "v194");
    asm volatile("v_mov_b32 v195, 0x7fc0beef" ::: "v195");
    asm volatile("v_mov_b32 v196, 0x7fc0beef" ::: "v196");
    asm volatile("v_mov_b32 v197, 0x7fc0beef" ::: "v197");
    asm volatile("v_mov_b32 v198, 0x7fc0beef" ::: "v198");
    asm volatile("v_mov_b32 v199, 0x7fc0beef" ::: "v199");
    asm volatile("v_mov_b32 v200, 0x7fc0beef" ::: "v200");
    asm volatile("v_mov_b32 v201, 0x7fc0beef" ::: "v201");
    asm volatile("v_mov_b32 v202, 0x7fc0beef" ::: "v202");
    asm volatile("v_mov_b32 v203, 0x7fc0beef" ::: "v203");
    asm volatile("v_mov_b32 v204, 0x7fc0beef" ::: "v204");
    asm volatile("v_mov_b32 v205, 0x7fc0beef" ::: "v205");
    asm volatile("v_mov_b32 v206, 0x7fc0beef" ::: "v206");
    asm volatile("v_mov_b32 v207, 0x7fc0beef" ::: "v207");
    asm volatile("v_mov_b32 v208, 0x7fc0beef" ::: "v208");
    asm volatile("v_mov_b32 v209, 0x7fc0beef" ::: "v209");
    asm volatile("v_mov_b32 v210, 0x7fc0beef" ::: "v210");
    asm volatile("v_mov_b32 v211, 0x7fc0beef" ::: "v211");
    asm volatile("v_mov_b32 v212, 0x7fc0beef" ::: "v212");
    asm volatile("v_mov_b32 v213, 0x7fc0beef" ::: "v213");
    asm volatile("v_mov_b32 v214, 0x7fc0beef" ::: "v214");
    asm volatile("v_mov_b32 v215, 0x7fc0beef" ::: "v215");
    asm volatile("v_mov_b32 v216, 0x7fc0beef" ::: "v216");
    asm volatile("v_mov_b32 v217, 0x7fc0beef" ::: "v217");
    asm volatile("v_mov_b32 v218, 0x7fc0beef" ::: "v218");
    asm volatile("v_mov_b32 v219, 0x7fc0beef" ::: "v219");
    asm volatile("v_mov_b32 v220, 0x7fc0beef" ::: "v220");
    asm volatile("v_mov_b32 v221, 0x7fc0beef" ::: "v221");
    asm volatile("v_mov_b32 v222, 0x7fc0beef" ::: "v222");
    asm volatile("v_mov_b32 v223, 0x7fc0beef" ::: "v223");
    asm volatile("v_mov_b32 v224, 0x7fc0beef" ::: "v224");
    asm volatile("v_mov_b32 v225, 0x7fc0beef" ::: "v225");
    asm volatile("v_mov_b32 v226, 0x7fc0beef" ::: "v226");
    asm volatile("v_mov_b32 v227, 0x7fc0beef" ::: "v227");
    asm volatile("v_mov_b32 v228, 0x7fc0beef" ::: "v228");
    asm volatile("v_mov_b32 v229, 0x7fc0beef" ::: "v229");
    asm volatile("v_mov_b32 v230, 0x7fc0beef" ::: "v230");
    asm volatile("v_mov_b32 v231, 0x7fc0beef" ::: "v231");
    asm volatile("v_mov_b32 v232, 0x7fc0beef" ::: "v232");
    asm volatile("v_mov_b32 v233, 0x7fc0beef" ::: "v233");
    asm volatile("v_mov_b32 v234, 0x7fc0beef" ::: "v234");
    asm volatile("v_mov_b32 v235, 0x7fc0beef" ::: "v235");
    asm volatile("v_mov_b32 v236, 0x7fc0beef" ::: "v236");
    asm volatile("v_mov_b32 v237, 0x7fc0beef" ::: "v237");
    asm volatile("v_mov_b32 v238, 0x7fc0beef" ::: "v238");
    asm volatile("v_mov_b32 v239, 0x7fc0beef" ::: "v239");
    asm volatile("v_mov_b32 v240, 0x7fc0beef" ::: "v240");
    asm volatile("v_mov_b32 v241, 0x7fc0beef" ::: "v241");
    asm volatile("v_mov_b32 v242, 0x7fc0beef" ::: "v242");
    asm volatile("v_mov_b32 v243, 0x7fc0beef" ::: "v243");
    asm volatile("v_mov_b32 v244, 0x7fc0beef" ::: "v244");
    asm volatile("v_mov_b32 v245, 0x7fc0beef" ::: "v245");
    asm volatile("v_mov_b32 v246, 0x7fc0beef" ::: "v246");
    asm volatile("v_mov_b32 v247, 0x7fc0beef" ::: "v247");
    asm volatile("v_mov_b32 v248, 0x7fc0beef" ::: "v248");
    asm volatile("v_mov_b32 v249, 0x7fc0beef" ::: "v249");
    asm volatile("v_mov_b32 v250, 0x7fc0beef" ::: "v250");
    asm volatile("v_mov_b32 v251, 0x7fc0beef" ::: "v251");
    asm volatile("v_mov_b32 v252, 0x7fc0beef" ::: "v252");
    asm volatile("v_mov_b32 v253, 0x7fc0beef" ::: "v253");
    asm volatile("v_mov_b32 v254, 0x7fc0beef" ::: "v254");
    asm volatile("v_mov_b32 v255, 0x7fc0beef" ::: "v255");
    if (seed == 12345.678f) sink[0] = seed;
}

// one-wave blocks, like the coder's kernels: a dependent integer / FP64 chain (the serial cores) ...
__global__ __launch_bounds__(64) void lk_chain1(float* sink, int iters, float seed) {
    unsigned int a = threadIdx.x * 2654435761u + blockIdx.x, b = 0x9E3779B9u;
    double p = 0.3 + 1e-3 * threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned int t = (unsigned int)(p * (double)(a >> 8));
            a = (a ^ (t << 3)) + b;
            b = (b << 1) | (b >> 31);
            a = __builtin_clz(a | 1u) + (a << 2);
        }
    }
    if (a == 12345u) sink[0] = (float)a;
}
// the same chain without any FP64 instruction (integers only) ...
__global__ __launch_bounds__(64) void lk_chain_int(float* sink, int iters, float seed) {
    unsigned int a = threadIdx.x * 2654435761u + blockIdx.x, b = 0x9E3779B9u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned int t = __umulhi(a, b) + (a >> 8);
            a = (a ^ (t << 3)) + b;
            b = (b << 1) | (b >> 31);
            a = __builtin_clz(a | 1u) + (a << 2);
        }
    }
    if (a == 12345u) sink[0] = (float)a;
}
// ... with the product formed by 64-bit integer multiply-adds (v_mad_u64_u32) instead of v_mul_f64 ...
__global__ __launch_bounds__(64) void lk_chain_mad64(float* sink, int iters, float seed) {
    unsigned int a = threadIdx.x * 2654435761u + blockIdx.x, b = 0x9E3779B9u;
    const unsigned int m_lo = 0x89ABCDEFu + threadIdx.x, m_hi = 0x00123456u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned long long lo = (unsigned long long)m_lo * (a >> 16);
            const unsigned long long hi = (unsigned long long)m_hi * (a >> 16) + (lo >> 32);
            const unsigned int t = (unsigned int)(hi >> 5);
            a = (a ^ (t << 3)) + b;
            b = (b << 1) | (b >> 31);
            a = __builtin_clz(a | 1u) + (a << 2);
        }
    }
    if (a == 12345u) sink[0] = (float)a;
}
// ... and FP64 only (convert, multiply, convert back: the three FP64 instructions of a coder step, nothing else)
__global__ __launch_bounds__(64) void lk_chain_f64(float* sink, int iters, float seed) {
    unsigned int a = threadIdx.x * 2654435761u + blockIdx.x;
    double p = 0.3 + 1e-3 * threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a = (unsigned int)(p * (double)(a >> 8)) + 0x9E3779B9u;
    }
    if (a == 12345u) sink[0] = (float)a;
}
// ... and a streaming pass over the block's own slice of a big buffer (the data-parallel passes: 1-4 bytes per decision)
__global__ __launch_bounds__(64) void lk_stream1(const float4* __restrict__ src, float4* __restrict__ dst, size_t per_block) {
    const size_t base = (size_t)blockIdx.x * per_block;
    for (size_t i = threadIdx.x; i < per_block; i += 64) dst[base + i] = src[base + i];
}

extern "C" int lk_launch(int kind, int blocks, int iters, void* buf, size_t bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* f = (float*)buf;
    switch (kind) {
    case 0: hipLaunchKernelGGL(lk_mfma, dim3(blocks), dim3(256), 0, s, f, iters, 1.f); break;
    case 1: hipLaunchKernelGGL(lk_valu, dim3(blocks), dim3(256), 0, s, f, iters, 1.f); break;
    case 2: hipLaunchKernelGGL(lk_lds, dim3(blocks), dim3(256), 0, s, f, iters, 1.f); break;
    case 3: hipLaunchKernelGGL(lk_mem, dim3(blocks), dim3(256), 0, s, (const float4*)buf, (float4*)((char*)buf + bytes / 2), bytes / 32); break;
    case 4: hipLaunchKernelGGL(lk_lds_dirty, dim3(blocks), dim3(256), 0, s, f, 0xDEADBEEFu); break;
    case 5: hipLaunchKernelGGL(lk_vgpr_dirty, dim3(blocks), dim3(64), 0, s, f, 1.f); break;
    case 6: hipLaunchKernelGGL(lk_chain1, dim3(blocks), dim3(64), 0, s, f, iters, 1.f); break;
    case 8: hipLaunchKernelGGL(lk_chain_int, dim3(blocks), dim3(64), 0, s, f, iters, 1.f); break;
    case 9: hipLaunchKernelGGL(lk_chain_mad64, dim3(blocks), dim3(64), 0, s, f, iters, 1.f); break;
    case 10: hipLaunchKernelGGL(lk_chain_f64, dim3(blocks), dim3(64), 0, s, f, iters, 1.f); break;
    case 7: hipLaunchKernelGGL(lk_stream1, dim3(blocks), dim3(64), 0, s, (const float4*)buf, (float4*)((char*)buf + bytes / 2), (size_t)iters); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
