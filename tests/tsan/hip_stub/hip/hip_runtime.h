// A STUB of the HIP runtime for tests/tsan ONLY: just enough of the API for the host files of libh263mi (batch.cpp,
// batch_staging.cpp, mixed_set.cpp, state.cpp, device_util.cpp) to compile with g++ -fsanitize=thread and run WITHOUT a GPU.
// "Device" and pinned memory are malloc; copies are memcpy executed at once on the calling thread (what a DMA engine would
// read is read here, so ThreadSanitizer sees it); streams and events do nothing; kernel launches are stubs (stub_runtime.cpp).
// Nothing here is a product path: the product is compiled by hipcc against the real runtime and has no CPU fallback.
#pragma once

#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotInitialized = 3,
    hipErrorInsufficientDriver = 35,
    hipErrorNoDevice = 100,
    hipErrorInvalidDevice = 101,
    hipErrorUnknown = 999,
} hipError_t;
typedef struct stub_stream *hipStream_t;
typedef struct stub_event *hipEvent_t;
typedef void *hipDeviceptr_t;
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
enum : unsigned {
    hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2,
    hipHostRegisterPortable = 1, hipHostRegisterMapped = 2,
    hipEventDisableTiming = 2, hipStreamNonBlocking = 1,
};

hipError_t hipGetDeviceCount(int *count);
hipError_t hipGetDevice(int *dev);
hipError_t hipSetDevice(int dev);
hipError_t hipGetLastError();
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetPCIBusId(char *id, int len, int dev);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t s);
hipError_t hipMemset(void *p, int v, size_t bytes);
hipError_t hipMemsetAsync(void *p, int v, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
