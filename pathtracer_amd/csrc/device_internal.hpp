// device_internal.hpp — what ptx_multi.hip needs of a device scene (ptx_device.hip owns the type).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/ptx.h"

namespace ptx {

// ptx_render_device with the progress callback of ptx_render; `stream` is a hipStream_t.
int internal_render(ptx_scene* scene, const ptx_pinhole* camera, uint64_t seed, uint32_t first_sample,
                    uint32_t sample_count, float* d_accum, void* stream, ptx_progress_cb progress, void* user);
// HIP device the scene lives on.
int internal_device(const ptx_scene* scene);
// Scene-owned buffers of a sharded render: this shard's accumulation buffer, (optionally) a landing buffer for
// a peer's, both of n_floats floats, and a non-blocking stream; grown on demand, freed with the scene.
int internal_multi_slot(ptx_scene* scene, size_t n_floats, float** accum, float** tmp, void** stream);

}  // namespace ptx
