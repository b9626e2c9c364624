"""Generates tools/microbench/kstep_asm.hip: the tile-split forward kernel's k-step (per wave 4 x 1 KB weight fragments through a buffer
descriptor, 4 x 1 KB activation fragments from LDS, 12 MFMAs) as ONE hand-scheduled inline-asm loop with named registers, in variants:

  where the loaded fragments live   arch VGPRs (what hipcc does) | accumulation registers (AGPRs: the loads' write traffic and the MFMAs'
                                    C / D traffic then sit in different halves of the register file)
  where the accumulators live       arch VGPRs | AGPRs
  issue order                       burst (4 loads, 4 reads, wait, 12 MFMAs) | interleaved (load, read, 3 MFMAs) x 4
  MFMA order                        chain (the three products of one accumulator back to back) | spread (product-major)

and a compiler-scheduled loop of the same arithmetic as the reference (every variant must reproduce its accumulator sums bit for bit).

    python tools/microbench/gen_kstep_asm.py && hipcc --offload-arch=gfx950 -O3 -o tools/microbench/kstep_asm tools/microbench/kstep_asm.hip
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def reg(file_, base, n=4):
    return '%s[%d:%d]' % (file_, base, base + n - 1)


def variant(name, ops_agpr, acc_agpr, interleave, spread, nloads=None, nreads=None, NT=2, NP=2, waves_per_simd=2):
    """NT tiles x NP column tiles per wave: 2 NT weight fragments (hi, lo per tile) and 2 NP activation fragments (hi, lo per column tile) per
    k-step, 3 NT NP MFMAs.  nloads / nreads below those counts: ablations (the remaining fragments keep their initial values)."""
    FA, FB = 2 * NT, 2 * NP
    nloads = FA if nloads is None else nloads
    nreads = FB if nreads is None else nreads
    n_ops, n_acc = (3 * FA + 2 * FB) * 4, NT * NP * 16
    of, af = ('a' if ops_agpr else 'v'), ('a' if acc_agpr else 'v')
    # register map
    if ops_agpr:
        a_base = 0                                   # a[0 ..]: ring of 3 x FA fragments, then two slots of FB
        acc_base = n_ops if acc_agpr else 32         # behind them | v[32 ..]
    else:
        a_base = 32 if acc_agpr else 32 + n_acc      # v[32 ..] | behind the accumulators
        acc_base = 0 if acc_agpr else 32             # a[0 ..] | v[32 ..]
    b_base = a_base + 3 * FA * 4
    A = lambda slot, f: reg(of, a_base + (slot * FA + f) * 4)
    B = lambda slot, f: reg(of, b_base + (slot * FB + f) * 4)
    ACC = lambda t, p: reg(af, acc_base + (t * NP + p) * 16, 16)
    step_lds = FB * 1024
    L = []
    emit = L.append

    def load(slot, f):
        if f < nloads:
            emit('buffer_load_dwordx4 %s, %%[voff], %%[rs], %%[so] offen offset:%d' % (A(slot, f), f * 1024))

    def advance():
        emit('s_add_u32 %%[so], %%[so], %d' % (FA * 1024))
        emit('s_cmp_gt_u32 %[so], %[wrap]')
        emit('s_cselect_b32 %[so], 0, %[so]')

    def read(slot, step, f):
        if f < nreads:
            emit('ds_read_b128 %s, %%[xl] offset:%d' % (B(slot, f), (step % 6) * step_lds + f * 1024))

    def mfma(t, p, prod, a, b):
        fa, fb = [(2 * t, 2 * p + 1), (2 * t + 1, 2 * p), (2 * t, 2 * p)][prod]
        emit('v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s' % (ACC(t, p), A(a, fa), B(b, fb), ACC(t, p)))

    def mfma_list():
        if spread:
            return [(t, p, prod) for prod in range(3) for t in range(NT) for p in range(NP)]
        return [(t, p, prod) for t in range(NT) for p in range(NP) for prod in range(3)]

    # initial values of every fragment register (the ablations multiply with them) and zero accumulators
    for i in range(n_ops):
        if ops_agpr:
            emit('v_accvgpr_write_b32 a%d, %%[one]' % (i,))
        else:
            emit('v_mov_b32 v%d, %%[one]' % (a_base + i,))
    for i in range(n_acc):
        if acc_agpr:
            emit('v_accvgpr_write_b32 a%d, 0' % (acc_base + i,))
        else:
            emit('v_mov_b32 v%d, 0' % (acc_base + i,))
    emit('s_mov_b32 %[so], 0')
    emit('s_nop 4')
    # prologue: weights of k-steps 0, 1 -> slots 0, 1; activations of k-step 0 -> slot 0
    for s in range(2):
        for f in range(FA):
            load(s, f)
        advance()
    for f in range(FB):
        read(0, 0, f)
    emit('1:')
    for u in range(6):
        nl, nr = nloads, nreads
        if interleave:                               # the k-step's loads and reads spread over its MFMAs, three MFMAs per slot
            order = mfma_list()
            slots = len(order) // 3
            for q in range(slots):
                fl = [f for f in range(FA) if f * slots // FA == q]
                fr = [f for f in range(FB) if f * slots // FB == q]
                for f in fl:
                    load((u + 2) % 3, f)
                for f in fr:
                    read((u + 1) & 1, u + 1, f)
                if q == 0:
                    emit('s_waitcnt vmcnt(%d) lgkmcnt(%d)' % (nl + len([f for f in fl if f < nl]), len([f for f in fr if f < nr])))
                for m in order[3 * q:3 * q + 3]:
                    mfma(*m, u % 3, u & 1)
            advance()
        else:
            for f in range(FA):
                load((u + 2) % 3, f)
            advance()
            for f in range(FB):
                read((u + 1) & 1, u + 1, f)
            emit('s_waitcnt vmcnt(%d) lgkmcnt(%d)' % (2 * nl, nr))
            for m in mfma_list():
                mfma(*m, u % 3, u & 1)
    emit('s_sub_u32 %[cnt], %[cnt], 1')
    emit('s_cmp_lg_u32 %[cnt], 0')
    emit('s_cbranch_scc1 1b')
    emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
    emit('s_nop 15')
    emit('s_nop 15')
    # the sum of the four accumulators' 64 registers, in the reference's order: r-major over (acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1])
    emit('v_mov_b32 %[sum], 0')
    for r in range(16):
        for tp in range(NT * NP):
            i = acc_base + tp * 16 + r
            if acc_agpr:
                emit('v_accvgpr_read_b32 %%[tmp], a%d' % i)
                emit('s_nop 1')
                src = '%[tmp]'
            else:
                src = 'v%d' % i
            if tp == 0:
                emit('v_mov_b32 %%[part], %s' % src)
            else:
                emit('v_add_f32 %%[part], %%[part], %s' % src)
        emit('v_add_f32 %[sum], %[sum], %[part]')
    clob = []
    clob += [('a%d' if ops_agpr else 'v%d') % (a_base + i) for i in range(n_ops)]
    clob += [('a%d' if acc_agpr else 'v%d') % (acc_base + i) for i in range(n_acc)]
    body = '\n'.join('        "%s\\n\\t"' % s for s in L)
    return '''
__global__ __launch_bounds__(256, %(wps)d) void %(name)s(const char* buf, long region_bytes, int regions, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const char* base = buf + (long)(blockIdx.x %% regions) * region_bytes + wave * (region_bytes / 4);
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(base)[i & 1023];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(region_bytes / 4), 0x00020000);
    const int voff = lane * 16;
    const unsigned xl = (unsigned)(size_t)lds + lane * 16;
    const int wrap = (int)(region_bytes / 4) - 8192;
    const unsigned one = 0x3c003c00u;
    float sum, part, tmp;
    int so, cnt = iters;
    const unsigned long long c0 = __builtin_readcyclecounter();
    asm volatile(
%(body)s
        : [sum] "=&v"(sum), [part] "=&v"(part), [tmp] "=&v"(tmp), [so] "=&s"(so), [cnt] "+s"(cnt)
        : [voff] "v"(voff), [xl] "v"(xl), [rs] "s"(rs), [wrap] "s"(wrap), [one] "v"(one)
        : "memory", "scc", %(clob)s);
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0) sink[threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(sink + 256)[0] = c1 - c0;
}
''' % dict(name=name, body=body, clob=', '.join('"%s"' % c for c in clob), wps=waves_per_simd)


HEAD = r'''// GENERATED by tools/microbench/gen_kstep_asm.py -- do not edit.  See that file for what the variants are.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// the compiler-scheduled loop (tools/microbench/kstep_loop.hip, PF = 3) on the generated variants' addresses: the reference
__global__ __launch_bounds__(256, 2) void k_compiler(const char* buf, long region_bytes, int regions, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const char* base = buf + (long)(blockIdx.x % regions) * region_bytes + wave * (region_bytes / 4);
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(base)[i & 1023];
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(region_bytes / 4), 0x00020000);
    const int voff = lane * 16;
    const char* xl = lds + lane * 16;
    f32x16 acc[2][2] = {{(f32x16)0.f, (f32x16)0.f}, {(f32x16)0.f, (f32x16)0.f}};
    u32x4 A[3][4];
    u32x4 B[2][4];
    int so = 0;
    const int wrap = (int)(region_bytes / 4) - 8192;
    const unsigned long long c0 = __builtin_readcyclecounter();
    auto loadA = [&](const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 4; ++f) A[slot][f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (f & 1) * 1024, so + (f >> 1) * 2048, 0));
        so += 4096; if (so > wrap) so = 0;
        asm volatile("" : "+s"(so));
    };
    loadA(0); loadA(1);
#pragma unroll
    for (int f = 0; f < 4; ++f) B[0][f] = *reinterpret_cast<const u32x4*>(xl + f * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            loadA((u + 2) % 3);
#pragma unroll
            for (int f = 0; f < 4; ++f) B[(u + 1) & 1][f] = *reinterpret_cast<const u32x4*>(xl + (((u + 1) % 6) * 4 + f) * 1024);
            const int a = u % 3, b = u & 1;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[t][p] = MF(A[a][2 * t], B[b][2 * p + 1], acc[t][p]);
                    acc[t][p] = MF(A[a][2 * t + 1], B[b][2 * p], acc[t][p]);
                    acc[t][p] = MF(A[a][2 * t], B[b][2 * p], acc[t][p]);
                }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += ((acc[0][0][r] + acc[0][1][r]) + acc[1][0][r]) + acc[1][1][r];
    asm volatile("" : "+v"(s));
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(sink + 256)[0] = c1 - c0;
}
'''

MAIN = r'''
typedef void (*kern_t)(const char*, long, int, int, float*);
static std::vector<float> ref;
static const char* only = nullptr;       // argv[1]: run only the variants whose description contains this text (the reference always runs) ...
static double soak_s = 0.0;              // ... argv[2]: and keep each of them running for this many seconds first (tools/clock_watch.py samples clocks / power)
static void run(kern_t kf, const char* buf, long region, int regions, float* sink, int cus, const char* what, bool check, int kb_loads, int kb_reads,
                int mfma_per_step = 12, int wgs_per_cu = 2) {
    const int iters = 278, grid = cus * wgs_per_cu * 4;       // 1668 k-steps per workgroup, four rounds of workgroups
    if (only && !strstr(what, only) && !ref.empty()) return;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(sink, 0, 256 * 4);
    if (only && strstr(what, only) && soak_s > 0.0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < soak_s) {
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, buf, region, regions, iters, sink);
            hipDeviceSynchronize();
        }
    }
    hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, buf, region, regions, 8, sink);
    hipDeviceSynchronize();
    std::vector<float> got(256);
    hipMemcpy(got.data(), sink, 256 * 4, hipMemcpyDeviceToHost);
    const char* verdict = "";
    if (ref.empty()) ref = got;
    else if (check) verdict = memcmp(ref.data(), got.data(), 256 * 4) == 0 ? "  == reference" : "  DIFFERS from the reference";
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, buf, region, regions, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, sink + 256, 8, hipMemcpyDeviceToHost);
    const double steps = 6.0 * iters, mfma = (double)grid * 4 * steps * mfma_per_step;
    // one wave's elapsed shader clocks against the matrix pipe's own time for the MFMAs of the waves that share its SIMD (32 cycles each);
    // the clock: a workgroup's share of the launch (grid / resident workgroups rounds) took that many cycles
    const double pipe = steps * mfma_per_step * 32.0 * wgs_per_cu / (double)cyc;
    const double ghz = (double)cyc * (grid / (double)(cus * wgs_per_cu)) / (best * 1e-3) * 1e-9;
    printf("%-82s %7.3f ms  %5.1f %% of 2.5 PF | pipe busy %5.1f %% of the wave's cycles, ~%4.2f GHz | %4.2f KB returned per MFMA (%d from L2 + %d from LDS per k-step)%s\n", what, best,
           mfma * 32768.0 / (best * 1e-3) / 2.5e15 * 100.0, pipe * 100.0, ghz, (kb_loads + kb_reads) / (double)mfma_per_step, kb_loads, kb_reads, verdict);
}

int main(int argc, char** argv) {
    if (argc > 1) only = argv[1];
    if (argc > 2) soak_s = atof(argv[2]);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const long region = 1638400;
    const int regions = 6;
    char* buf; float* sink;
    hipMalloc(&buf, region * regions);
    hipMalloc(&sink, 256 * 4 + 16);
    std::vector<unsigned short> h(region * regions / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00u + (i * 2654435761u >> 22 & 0x3ffu));
    hipMemcpy(buf, h.data(), region * regions, hipMemcpyHostToDevice);
    run(k_compiler, buf, region, regions, sink, cus, "compiler-scheduled loop (the kernel's k-step)", true, 4, 4);
%(runs)s
    return 0;
}
'''


def main():
    variants = []
    for ops_agpr in (0, 1):
        for acc_agpr in (0, 1):
            for inter in (0, 1):
                for spread in (0, 1):
                    name = 'k_ops%s_acc%s_%s_%s' % ('A' if ops_agpr else 'V', 'A' if acc_agpr else 'V', 'inter' if inter else 'burst', 'spread' if spread else 'chain')
                    what = 'asm: fragments in %s, accumulators in %s, %s, %s' % ('AGPRs' if ops_agpr else 'VGPRs', 'AGPRs' if acc_agpr else 'VGPRs',
                                                                                  'interleaved' if inter else 'burst', 'spread' if spread else 'chain')
                    variants.append((name, what, variant(name, ops_agpr, acc_agpr, inter, spread), True, 4, 4, 12, 2))
    # ablations on the two extreme register placements: no loads, no reads, neither
    for ops_agpr in (0, 1):
        for nl, nr, tag in ((0, 4, 'LDS reads only'), (4, 0, 'L2 loads only'), (0, 0, 'MFMAs only')):
            name = 'k_abl_ops%s_%d_%d' % ('A' if ops_agpr else 'V', nl, nr)
            what = 'asm ablation: %s, fragments in %s, accumulators in VGPRs' % (tag, 'AGPRs' if ops_agpr else 'VGPRs')
            variants.append((name, what, variant(name, ops_agpr, 0, 1, 1, nl, nr), False, nl, nr, 12, 2))
    # other register tiles (tiles x column tiles per wave); the larger ones need a 512-register wave: one wave per SIMD
    for NT, NP, wps, acc_agpr in ((1, 4, 2, 0), (2, 2, 1, 0), (2, 4, 1, 1), (4, 2, 1, 1), (4, 4, 1, 1)):
        name = 'k_tile_%dx%d_w%d' % (NT, NP, wps)
        what = 'asm: %d tile(s) x %d column tiles per wave, %d wave(s) per SIMD, interleaved, spread' % (NT, NP, wps)
        variants.append((name, what, variant(name, 0, acc_agpr, 1, 1, NT=NT, NP=NP, waves_per_simd=wps), False, 2 * NT, 2 * NP, 3 * NT * NP, wps))
    src = HEAD + ''.join(v[2] for v in variants)
    runs = '\n'.join('    run(%s, buf, region, regions, sink, cus, "%s", %s, %d, %d, %d, %d);' % (v[0], v[1], 'true' if v[3] else 'false', v[4], v[5], v[6], v[7])
                     for v in variants)
    src += MAIN.replace('%(runs)s', runs)
    with open(os.path.join(HERE, 'kstep_asm.hip'), 'w') as f:
        f.write(src)


if __name__ == '__main__':
    main()
