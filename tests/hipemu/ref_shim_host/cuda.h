#pragma once
#define CUDA_VERSION 12000
#include "cuda_runtime.h"
