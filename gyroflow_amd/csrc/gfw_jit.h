// gfw_jit.h — run-time specialisation of the fused frame kernel (gfw_jit.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "gfw_frame.h"

enum { GFW_JIT_UNAVAILABLE = 0, GFW_JIT_COMPILING = 1, GFW_JIT_READY = 2, GFW_JIT_FAILED = -1 };
struct GfwJitInfo { int state; double compile_ms; std::string log; };

bool gfw_jit_available();
std::string gfw_jit_source_id();
std::string gfw_jit_cache_name(const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header);
hipFunction_t gfw_jit_get(int device, const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header,
                          bool wait, GfwJitInfo *info);
long gfw_jit_compile_only(const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header, std::string &log,
                          std::vector<char> *code_out);
hipError_t gfw_jit_launch(hipFunction_t fn, const GfwClipArgs &C, int grid, hipStream_t s);
bool gfw_jit_write_code_object(const std::string &path, const std::vector<char> &code);
bool gfw_jit_read_symbol(hipFunction_t fn, const char *name, void *dst, size_t bytes);
