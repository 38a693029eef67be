// Error reporting shared by the translation units of the C-ABI library (gsv_last_error() lives in gsv_abi.hip).
#pragma once
namespace gsv {
__attribute__((visibility("hidden"))) int abi_fail(int code, const char* fmt, ...);
}
