// ref_knn_capi.cpp -- TEST INFRASTRUCTURE.  extern "C" wrapper around the REFERENCE's SimpleKNN::knn
// (reference submodules/simple-knn/simple_knn.h, simple_knn.cu:185-221) compiled for gfx950 by oracle/build_ref.sh from the
// sources under /root/reference.  Plays the role of the torch binding distCUDA2 (spatial.cu:15-26) without torch.
#include "cuda_runtime.h"
#include "simple_knn.h"

extern "C" int knnref_mean_dist3(int P, const float* points_dev, float* mean_dists_dev)
{
    if (P <= 0) return 0;
    if (hipMemset(mean_dists_dev, 0, sizeof(float) * (size_t)P) != hipSuccess) return -1;     // torch::full({P}, 0.0), spatial.cu:21
    SimpleKNN::knn(P, (float3*)points_dev, mean_dists_dev);
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
