// Helpers the translation units of libmorb_hip.so share with each other (handle fields, grow-only workspaces).  C linkage so that the
// handle structs stay private to their units, HIDDEN visibility so that the library exports exactly what include/morb_hip.h declares
// (tests/test_oracle_cpu.py compares `nm -D` with the header).
#pragma once
#include <cstddef>

struct morb_matcher;
struct morb_optimizer;
#define MORB_INTERNAL __attribute__((visibility("hidden")))
extern "C" {
MORB_INTERNAL int morb_matcher_device(const morb_matcher*);
MORB_INTERNAL int morb_matcher_workspace(morb_matcher*, int which, size_t bytes, void** out);
MORB_INTERNAL int morb_matcher_const(morb_matcher*, int slot, const void* host, size_t bytes, void** d_out, void* stream);
MORB_INTERNAL int morb_bow_sort_images(morb_matcher* m, int nimg, const int* d_node, const int* d_count, int cap, unsigned long long** d_sorted,
                                       void* stream);
MORB_INTERNAL int morb_optimizer_device(const morb_optimizer*);
MORB_INTERNAL int morb_optimizer_workspace(morb_optimizer*, size_t bytes, void** out);
MORB_INTERNAL int morb_optimizer_lm_words(morb_optimizer*, int** host, int** dev);   // 16 pinned, device-mapped ints (LM state mirror)
MORB_INTERNAL int morb_optimizer_staging(morb_optimizer*, size_t bytes, void** host);   // grow-only pinned host buffer
MORB_INTERNAL int morb_optimizer_spill(morb_optimizer*, size_t bytes, void** out);      // grow-only device buffer of the batch entry points (k_pose_inertial's edge lists beyond the LDS)
}
