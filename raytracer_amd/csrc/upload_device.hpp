// rta_upload_scene for large scenes, on the GPU: validation of the description,
// the kernels' layouts (pair, primitive, shading records) and the wide tree of the
// certified walk are made by kernels from the caller's arrays, copied to the
// device as they are (upload_device.hip).  The host-side path in context.cpp and
// wide_tree.cpp stays the one for small scenes and the one that words the error
// of an invalid description.
#pragma once

#include "context.hpp"

namespace rta {

enum {
	kUploadDone = 0,          // the scene is on the device, the context's scene fields are set
	kUploadNotHandled = 1     // nothing was changed: the host-side path has to do it (an invalid description, whose
	                          // error it words; a leaf too long for one thread to cut down; a tree deeper than the limit)
};

// Primitives from which on rta_upload_scene takes this path (RTA_UPLOAD_DEVICE=0 / 1 forces the choice,
// for tests and measurements).  It is the size beyond which the host-side builder of the wide tree keeps
// the reference's topology (kWideSahLimit), which is what this path does on the device.
bool upload_goes_to_device(const rta_scene_desc &d);

// Returns kUploadDone, kUploadNotHandled or a negative RTA_E_* (HIP failure; ctx->error is set).
// Materials and lights (a few records) are left to the caller.
int upload_scene_on_device(rta_context *ctx, const rta_scene_desc &d);

}  // namespace rta
