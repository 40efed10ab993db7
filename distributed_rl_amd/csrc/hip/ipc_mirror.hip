// Device buffers shared between PROCESSES on one GPU (hipIpc): the learner's parameter mirror (transport.ShmParams: actors on
// the learner's GPU copy the newest weights device to device, no PCIe) and raw stream-ordered copies into / out of them.
// The buffers come from hipMalloc directly (an IPC handle names a whole allocation, not a piece of a caching allocator's
// block).
#include "common.h"
#include <string.h>

static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 opaque bytes (drl_host.cpp MirrorRecord)");

DRL_API int drl_ipc_alloc(unsigned long long bytes, void** ptr_out, void* handle64_out) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return (int)e;
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    memcpy(handle64_out, &h, 64);
    *ptr_out = p;
    return 0;
}
DRL_API int drl_ipc_free(void* ptr) { return (int)hipFree(ptr); }
DRL_API int drl_ipc_open(const void* handle64, void** ptr_out) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    return (int)hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess);
}
DRL_API int drl_ipc_close(void* ptr) { return (int)hipIpcCloseMemHandle(ptr); }
DRL_API int drl_copy_d2d_async(void* dst, const void* src, unsigned long long bytes, hipStream_t s) {
    return (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
}
// PCI bus id of device `dev` ("0000:c1:00.0"): the same physical GPU has the same id in every process, whatever
// HIP_VISIBLE_DEVICES made of its index
DRL_API int drl_device_bus_id(int dev, char* out64) {
    memset(out64, 0, 64);
    return (int)hipDeviceGetPCIBusId(out64, 63, dev);
}
