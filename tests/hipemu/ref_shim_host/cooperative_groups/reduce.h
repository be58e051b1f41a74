#pragma once
#include "../cooperative_groups.h"
