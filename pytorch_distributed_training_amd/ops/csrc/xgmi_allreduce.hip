// Host side and the stand-alone kernel of the one-shot xGMI all-reduce (protocol: xgmi_exchange.h).
#include "xgmi_exchange.h"

__global__ __launch_bounds__(1024) void car_allreduce_kernel(const CarDev d, float* __restrict__ data, int n) {
    __shared__ float vals[CAR_MAX_FLOATS];
    __shared__ unsigned ctl[CAR_CTL_WORDS];
    for (int i = threadIdx.x; i < n; i += blockDim.x) vals[i] = data[i];
    __syncthreads();
    car_exchange_block(d, vals, n, ctl);            // after a timeout: NaN everywhere, status set
    for (int i = threadIdx.x; i < n; i += blockDim.x) data[i] = vals[i];
}

// Any length: the tensor in pieces of ``chunk`` floats, one exchange per piece (each advances the sequence number).  After a
// failed piece the remaining ones return at once (the status word is sticky), so the kernel always ends.
__global__ __launch_bounds__(1024) void car_allreduce_chunks_kernel(const CarDev d, float* __restrict__ data, long long n,
                                                                    int chunk) {
    __shared__ float vals[CAR_MAX_FLOATS];
    __shared__ unsigned ctl[CAR_CTL_WORDS];
    for (long long off = 0; off < n; off += chunk) {
        const int m = (int)(n - off < (long long)chunk ? n - off : (long long)chunk);
        for (int i = threadIdx.x; i < m; i += blockDim.x) vals[i] = data[off + i];
        __syncthreads();
        car_exchange_block(d, vals, m, ctl);
        for (int i = threadIdx.x; i < m; i += blockDim.x) data[off + i] = vals[i];
        __syncthreads();
    }
}

static size_t car_recv_offset() { return 4096; }   // flags live in the first page of the allocation

PDT_API int pdt_car_create(int world, int rank, int max_floats, void** out) {
    if (!out || world < 1 || world > CAR_MAX_WORLD || rank < 0 || rank >= world || max_floats < 4 || (max_floats & 3))
        return -2;
    CarHost* h = new CarHost();
    h->dev.world = world;
    h->dev.rank = rank;
    h->dev.max4 = max_floats / 4;
    h->local_bytes = car_recv_offset() + (size_t)2 * world * max_floats * sizeof(float);
    hipError_t e = hipExtMallocWithFlags(&h->local_base, h->local_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { delete h; return (int)e; }
    e = hipMemset(h->local_base, 0, h->local_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&h->seq_status, 4 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(h->seq_status, 0, 4 * sizeof(unsigned));
    void* pinned_dev = nullptr;
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->status_pinned, 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        *h->status_pinned = 0u;
        e = hipHostGetDevicePointer(&pinned_dev, h->status_pinned, 0);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(h->local_base);
        if (h->seq_status) (void)hipFree(h->seq_status);
        if (h->status_pinned) (void)hipHostFree(h->status_pinned);
        delete h;
        return (int)e;
    }
    h->dev.seq = h->seq_status;
    h->dev.status = h->seq_status + 1;
    h->dev.rereads = h->seq_status + 2;
    h->dev.status_host = static_cast<unsigned*>(pinned_dev);
    h->dev.flags = reinterpret_cast<unsigned*>(h->local_base);
    h->dev.recv = reinterpret_cast<float4*>(static_cast<char*>(h->local_base) + car_recv_offset());
    for (int r = 0; r < CAR_MAX_WORLD; ++r) {
        h->dev.peer_flags[r] = nullptr;
        h->dev.peer_recv[r] = nullptr;
        h->opened[r] = nullptr;
    }
    h->dev.peer_flags[rank] = h->dev.flags;
    h->dev.peer_recv[rank] = h->dev.recv;
    h->dev.timeout_ticks = 200000000LL;           // 2 s of the 100 MHz wall clock
    h->dev.corrupt = 0u;
    *out = h;
    return 0;
}

// 64-byte IPC handle of the local buffer, to be sent to the peers.
PDT_API int pdt_car_ipc_handle(void* handle, void* out64) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !out64) return -2;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    return (int)hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(out64), h->local_base);
}

// handles: world x 64 bytes, indexed by rank (the own entry is ignored).
PDT_API int pdt_car_open(void* handle, const void* handles) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !handles) return -2;
    const hipIpcMemHandle_t* hs = static_cast<const hipIpcMemHandle_t*>(handles);
    for (int r = 0; r < h->dev.world; ++r) {
        if (r == h->dev.rank) continue;
        void* base = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&base, hs[r], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return (int)e;
        h->opened[r] = base;
        h->dev.peer_flags[r] = reinterpret_cast<unsigned*>(base);
        h->dev.peer_recv[r] = reinterpret_cast<float4*>(static_cast<char*>(base) + car_recv_offset());
    }
    return 0;
}

PDT_API int pdt_car_set_timeout_ms(void* handle, int ms) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || ms < 1) return -2;
    h->dev.timeout_ticks = (long long)ms * 100000LL;
    return 0;
}

// Test hook: from now on this rank sends checksums XORed with ``mask`` (0: back to normal).  A receiver must then refuse
// the payload (status 2, NaN result) after its bounded re-reads.
PDT_API int pdt_car_corrupt_checksum(void* handle, unsigned mask) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h) return -2;
    h->dev.corrupt = mask;
    return 0;
}

// In-place sum over the ranks of data[0..n): n a multiple of 4, data 16-byte aligned.
PDT_API int pdt_car_allreduce(void* handle, float* data, int n, hipStream_t stream) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !data || n < 4 || (n & 3) || n / 4 > h->dev.max4 || n > CAR_MAX_FLOATS || (reinterpret_cast<uintptr_t>(data) & 15)) return -2;
    for (int r = 0; r < h->dev.world; ++r)
        if (!h->dev.peer_recv[r]) return -3;       // peers not opened yet
    hipLaunchKernelGGL(car_allreduce_kernel, dim3(1), dim3(1024), 0, stream, h->dev, data, n);
    return pdt_launch_status();
}

PDT_API int pdt_car_allreduce_chunks(void* handle, float* data, long long n, hipStream_t stream) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !data || n < 1 || (reinterpret_cast<uintptr_t>(data) & 3)) return -2;
    for (int r = 0; r < h->dev.world; ++r)
        if (!h->dev.peer_recv[r]) return -3;
    const int cap = h->dev.max4 * 4;
    const int chunk = cap < CAR_MAX_FLOATS ? cap : CAR_MAX_FLOATS;
    hipLaunchKernelGGL(car_allreduce_chunks_kernel, dim3(1), dim3(1024), 0, stream, h->dev, data, n, chunk);
    return pdt_launch_status();
}

// Synchronous read of [sequence number, status, payload re-reads] (status != 0: a wait timed out or a checksum failed).
PDT_API int pdt_car_query(void* handle, unsigned* out3) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !out3) return -2;
    return (int)hipMemcpy(out3, h->seq_status, 3 * sizeof(unsigned), hipMemcpyDeviceToHost);
}

// Status word as last written to pinned host memory: no synchronisation, no GPU call.  May lag
// the device by the kernels still in flight; it never goes back to zero.
PDT_API int pdt_car_peek(void* handle) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h || !h->status_pinned) return -2;
    return (int)*static_cast<volatile unsigned*>(h->status_pinned);
}

PDT_API int pdt_car_destroy(void* handle) {
    CarHost* h = static_cast<CarHost*>(handle);
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < CAR_MAX_WORLD; ++r)
        if (h->opened[r]) (void)hipIpcCloseMemHandle(h->opened[r]);
    (void)hipFree(h->local_base);
    (void)hipFree(h->seq_status);
    if (h->status_pinned) (void)hipHostFree(h->status_pinned);
    delete h;
    return 0;
}
