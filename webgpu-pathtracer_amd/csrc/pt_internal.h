// pt_internal.h -- small helpers shared by the translation units of libmi3pt.so
#pragma once
#include <string>

// Records `msg` as the calling thread's last error and returns `code`.
int pt_set_error(int code, const std::string &msg);
