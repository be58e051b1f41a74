// mtets.hip -- marching tetrahedra (replaces utils/tetmesh.py:47-138). Implemented below.
#include "gof_common.h"
extern "C" {
size_t gof_mtets_ws_bytes(int64_t num_tets) { (void)num_tets; return 0; }
int gof_mtets_count(int64_t, int64_t, const int64_t*, const float*, void*, size_t, int64_t*, int64_t*, void*) { gof::set_error("mtets: not implemented yet"); return GOF_E_INVALID; }
int gof_mtets_emit(int64_t, int64_t, const int64_t*, const float*, const float*, const float*, const void*, size_t, int64_t, int64_t,
                   int64_t*, float*, float*, float*, int64_t*, void*) { gof::set_error("mtets: not implemented yet"); return GOF_E_INVALID; }
}
