// api_common.h -- shared by the translation units of libdctts_hip.so.
#pragma once
#include <string>

// Records `msg` for dctts_last_error() (thread-local) and returns `code`.  Defined in dctts_api.hip.
int dctts_set_error(int code, const std::string& msg);
