// gs_merge.hip -- merge of the accumulators of several match runs that live in ONE process (a JVM host has no
// torch.distributed): the runs of one device are reduced by small kernels, the devices among each other by RCCL
// collectives over xGMI (all-reduce SUM / MAX, all-gather + OR for the unique-k-mer bitmap -- RCCL has no bitwise OR),
// exactly the algebra of genestrip_amd/distributed.py: merge_run_state.  RCCL is resolved at run
// time (dlopen), so that hosts with one GPU need no librccl and a process that already holds a copy (PyTorch) is not
// given a second one at load time.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gs_params.h"

typedef unsigned long long u64;

__global__ __launch_bounds__(256) void gs_merge_i64_kernel(long long *dst, const long long *src, int64_t n, int op) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long a = dst[i], b = src[i];
        dst[i] = op == 0 ? a + b : (a > b ? a : b);
    }
}

__global__ __launch_bounds__(256) void gs_merge_f64_kernel(double *dst, const double *src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

static int merge_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

// dst op= src on the current device (op 0: sum, 1: max)
extern "C" hipError_t gs_launch_merge_i64(void *dst, const void *src, int64_t n, int op, hipStream_t stream) {
    hipLaunchKernelGGL(gs_merge_i64_kernel, dim3(merge_grid(n)), dim3(256), 0, stream, (long long *)dst, (const long long *)src, n, op);
    return hipGetLastError();
}
extern "C" hipError_t gs_launch_merge_f64(void *dst, const void *src, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(gs_merge_f64_kernel, dim3(merge_grid(n)), dim3(256), 0, stream, (double *)dst, (const double *)src, n);
    return hipGetLastError();
}

// ---- RCCL, bound at run time
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            error = "librccl.so not found (needed to merge runs of different devices)";
            return false;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(lib, n);
            if (!p) {
                ok = false;
                error = std::string("librccl lacks ") + n;
            }
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(lib);
            lib = nullptr;
        }
        return ok;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;  // one communicator set per device list, kept for the process
}  // namespace

// One leader buffer set per device, all of the same shape:
//   sums int64[n_sums] (SUM)   maxk int64[n_max] (MAX)   dsums double[n_dsums] (SUM)   bitmap uint32[n_words] (OR)
// gather[i]: device scratch of n_dev * n_words uint32 on device i (the all-gathered bitmaps).  Afterwards every leader
// holds the global state.  Returns 0 or a negative code; *msg describes a failure.
extern "C" int gs_rccl_merge_leaders(int n_dev, const int *devices, void *const *sums, void *const *maxk, void *const *dsums,
                                     void *const *bitmap, void *const *gather, int64_t n_sums, int64_t n_max, int64_t n_dsums,
                                     int64_t n_words, const hipStream_t *streams, const char **msg) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    static std::string err;
    if (!g_rccl.load()) {
        err = g_rccl.error;
        *msg = err.c_str();
        return -4;
    }
    std::vector<int> key(devices, devices + n_dev);
    auto it = g_comms.find(key);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> comms((size_t)n_dev);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_dev, devices);
        if (r != ncclSuccess) {
            err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r);
            *msg = err.c_str();
            return -3;
        }
        it = g_comms.emplace(key, comms).first;
    }
    const std::vector<ncclComm_t> &comms = it->second;
    ncclResult_t r = ncclSuccess;
    auto step = [&](ncclResult_t x) {
        if (r == ncclSuccess) r = x;
    };
    step(g_rccl.GroupStart());
    for (int i = 0; i < n_dev; i++) {
        hipSetDevice(devices[i]);
        step(g_rccl.AllReduce(sums[i], sums[i], (size_t)n_sums, ncclInt64, ncclSum, comms[(size_t)i], streams[i]));
        step(g_rccl.AllReduce(maxk[i], maxk[i], (size_t)n_max, ncclInt64, ncclMax, comms[(size_t)i], streams[i]));
        step(g_rccl.AllReduce(dsums[i], dsums[i], (size_t)n_dsums, ncclDouble, ncclSum, comms[(size_t)i], streams[i]));
        if (n_words > 0)
            step(g_rccl.AllGather(bitmap[i], gather[i], (size_t)n_words, ncclUint32, comms[(size_t)i], streams[i]));
    }
    step(g_rccl.GroupEnd());
    if (r != ncclSuccess) {
        err = std::string("RCCL merge: ") + g_rccl.GetErrorString(r);
        *msg = err.c_str();
        return -3;
    }
    return 0;
}
