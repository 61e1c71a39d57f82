// cumask_probe.hip -- feasibility probe (measurement tool, not part of the product): can two PROCESSES on one MI355X each keep a
// resident kernel on a disjoint set of CUs (hipExtStreamCreateWithCUMask) and talk through IPC-mapped device memory while both run?
// That is what an in-kernel neighbour exchange between shards needs; RCCL refuses two ranks on one device, so
// on the 1-GPU boxes this is the only way to exercise it.
//
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o gpurun_out/cumask_probe && gpurun_out/cumask_probe
//
// Parent and child each: allocate an uncached word pair, exchange IPC handles over a pipe, open the peer's, create a stream masked to
// one half of the CUs, launch `blocks` workgroups of a ping-pong kernel: rank r waits until the peer's counter reaches k, then
// raises its own to k + 1 (system-scope atomics), `rounds` times.  Prints the rounds per second and the per-round latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", (int)getpid(), #x, hipGetErrorString(e_)); _exit(2); } } while (0)

__global__ void __launch_bounds__(256) pingpong(unsigned long long *mine, const unsigned long long *peer, int rank, int rounds, unsigned long long *out,
                                                unsigned long long spin_limit) {
    // every workgroup spins (as a resident engine kernel would); workgroup 0 plays the ping-pong
    __shared__ int stop;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    unsigned long long spins = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        for (int k = 0; k < rounds; ++k) {
            const unsigned long long want = (unsigned long long)(2 * k + rank);       // rank 0 starts: 0 -> 1, rank 1: 1 -> 2, ...
            while (__hip_atomic_load(peer, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
                if (++spins > spin_limit) { out[2] = 1; k = rounds; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __hip_atomic_store(mine, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        out[0] = wall_clock64() - t0;
        out[1] = spins;
        __hip_atomic_store(mine + 1, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // done
    } else if (threadIdx.x == 0) {
        while (__hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
            if (++spins > spin_limit / 8) break;
            __builtin_amdgcn_s_sleep(32);
        }
    }
}

static int run(int rank, int rfd, int wfd, int use_mask, int blocks, int rounds) {
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *mine = nullptr, *out = nullptr;
    CK(hipExtMallocWithFlags((void **)&mine, 4096, hipDeviceMallocUncached));
    CK(hipMemset(mine, 0, 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(out, 0, 64));
    hipIpcMemHandle_t hm, hp;
    CK(hipIpcGetMemHandle(&hm, mine));
    if (write(wfd, &hm, sizeof hm) != (ssize_t)sizeof hm) return 3;
    if (read(rfd, &hp, sizeof hp) != (ssize_t)sizeof hp) return 3;
    unsigned long long *peer = nullptr;
    CK(hipIpcOpenMemHandle((void **)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    hipStream_t st;
    if (use_mask) {
        // half of the CUs of EVERY XCD: bit i of the mask = CU i of the device's linear CU numbering
        std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
        for (int i = 0; i < n_cu; ++i) if ((i & 1) == rank) mask[i / 32] |= 1u << (i % 32);
        CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    } else CK(hipStreamCreate(&st));
    char go = 'g';                                   // both are set up: start together
    if (write(wfd, &go, 1) != 1 || read(rfd, &go, 1) != 1) return 3;
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(pingpong, dim3(blocks), dim3(256), 0, st, mine, peer, rank, rounds, out, 20000000ull);      // (bounded spins: ~2 s, never a hang)
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long o[3];
    CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
    printf("{\"rank\": %d, \"cu_mask\": %d, \"blocks\": %d, \"rounds\": %d, \"wall_s\": %.4f, \"device_ticks_100MHz\": %llu, \"spins\": %llu, \"gave_up\": %llu, "
           "\"us_per_round_trip\": %.3f}\n", rank, use_mask, blocks, rounds, wall, o[0], o[1], o[2], (double)o[0] / 100.0 / rounds * 2.0);
    fflush(stdout);
    CK(hipIpcCloseMemHandle(peer));
    return o[2] ? 4 : 0;
}

int main(int argc, char **argv) {
    const int use_mask = argc > 1 ? atoi(argv[1]) : 1, blocks = argc > 2 ? atoi(argv[2]) : 128, rounds = argc > 3 ? atoi(argv[3]) : 2000;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    const pid_t pid = fork();                         // (before any HIP call: each process initialises the runtime itself)
    if (pid == 0) { close(p2c[1]); close(c2p[0]); _exit(run(1, p2c[0], c2p[1], use_mask, blocks, rounds)); }
    close(p2c[0]); close(c2p[1]);
    const int rc = run(0, c2p[0], p2c[1], use_mask, blocks, rounds);
    int status = 0;
    waitpid(pid, &status, 0);
    return rc ? rc : (WIFEXITED(status) ? WEXITSTATUS(status) : 5);
}
