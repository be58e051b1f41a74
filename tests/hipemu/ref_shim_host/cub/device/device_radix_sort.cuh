#pragma once
#include "../cub.cuh"
