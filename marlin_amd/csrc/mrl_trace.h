#pragma once
#include <cstdio>

// debugging aid (MRL_OPT_EXPERIMENT bit 1 << 20): host-side progress lines of the multi-GPU drivers on stderr
extern int g_mrl_trace;
#define MRL_TRACE(...)                                   \
  do {                                                   \
    if (g_mrl_trace) {                                   \
      std::fprintf(stderr, "[mrl trace] " __VA_ARGS__);  \
      std::fprintf(stderr, "\n");                        \
      std::fflush(stderr);                               \
    }                                                    \
  } while (0)

