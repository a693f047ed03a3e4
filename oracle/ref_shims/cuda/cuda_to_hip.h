// Force-included when the reference's CUDA/Thrust matcher sources (slam/thirdparty/fast_gicp/src/fast_gicp/cuda/*.cu) are compiled for
// gfx950 as TEST INFRASTRUCTURE (oracle/ref_ndt_cuda.hip): the handful of CUDA runtime names those files use, spelled in HIP.  rocThrust
// provides the thrust:: API they are written against.  This is not a compatibility layer of the product -- nothing under
// lidar-slam-detection_amd/ includes it.
#pragma once
#include <hip/hip_runtime.h>
typedef hipStream_t cudaStream_t;
#define cudaStreamNonBlocking hipStreamNonBlocking
#define cudaStreamCreateWithFlags hipStreamCreateWithFlags
#define cudaStreamSynchronize hipStreamSynchronize
#define cudaStreamDestroy hipStreamDestroy
#define cudaDeviceSynchronize hipDeviceSynchronize
// rocThrust spells the device back end thrust::hip / thrust::system::hip; the sources say thrust::cuda::par.on(stream) and
// thrust::system::cuda::unique_eager_event
#include <thrust/execution_policy.h>
#include <thrust/system/hip/execution_policy.h>
#include <thrust/system/hip/future.h>
namespace thrust {
namespace cuda = ::thrust::hip;
namespace system {
namespace cuda = ::thrust::system::hip;
}
}  // namespace thrust
