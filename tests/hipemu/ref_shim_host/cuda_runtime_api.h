#pragma once
#include "cuda_runtime.h"
