// svsx_search.hip -- dispatch of the batched greedy search to the per-row-type translation units (svsx_search_kernel.h holds the
// kernel; svsx_search_{f32,f16,i8,u8}.hip its instantiations).
#include "svsx_device.h"
#include "svsx_kernels.h"

namespace svsx {

static thread_local char g_last_inst[96] = "";
const char* last_search_instantiation() { return g_last_inst; }
char* last_search_instantiation_buffer() { return g_last_inst; }

hipError_t launch_search_f32(int metric, int qmode, const SearchArgs& a, hipStream_t stream);
hipError_t launch_search_f16(int metric, int qmode, const SearchArgs& a, hipStream_t stream);
hipError_t launch_search_i8(int metric, int qmode, const SearchArgs& a, hipStream_t stream);
hipError_t launch_search_u8(int metric, int qmode, const SearchArgs& a, hipStream_t stream);

hipError_t launch_search(int data_dtype, int metric, int qmode, const SearchArgs& a, hipStream_t stream) {
    switch (data_dtype) {
        case DT_F32: return launch_search_f32(metric, qmode, a, stream);
        case DT_F16: return launch_search_f16(metric, qmode, a, stream);
        case DT_I8: return launch_search_i8(metric, qmode, a, stream);
        case DT_U8: return launch_search_u8(metric, qmode, a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace svsx
