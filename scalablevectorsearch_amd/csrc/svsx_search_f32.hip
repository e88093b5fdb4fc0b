// svsx_search_f32.hip -- the search_kernel instantiations for f32 rows (svsx_search_kernel.h); one translation unit per row type so
// that they compile in parallel.
#include "svsx_search_kernel.h"

namespace svsx {

hipError_t launch_search_f32(int metric, int qmode, const SearchArgs& a, hipStream_t stream) { return launch_search_for_rows<DT_F32>(metric, qmode, a, stream); }

} // namespace svsx
