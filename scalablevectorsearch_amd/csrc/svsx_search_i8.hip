// svsx_search_i8.hip -- the search_kernel instantiations for i8 rows (svsx_search_kernel.h); one translation unit per row type so
// that they compile in parallel.
#include "svsx_search_kernel.h"

namespace svsx {

hipError_t launch_search_i8(int metric, int qmode, const SearchArgs& a, hipStream_t stream) { return launch_search_for_rows<DT_I8>(metric, qmode, a, stream); }

} // namespace svsx
