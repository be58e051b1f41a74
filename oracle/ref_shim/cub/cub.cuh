#pragma once
#include <cstring>
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
