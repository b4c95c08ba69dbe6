// k_project.hip -- Frame grid + ORBmatcher::SearchByProjection(Frame&, MapPoints, th)
// (Frame.cc:461-476,593-658; ORBmatcher.cc:155-249).  Placeholder until the device kernels
// land: fails loudly, never falls back to a CPU path.
#include "gfo_internal.h"

extern "C" int gfo_search_by_projection(gfo_ctx* c, const gfo_keypoint*, const uint8_t*, const float*, int,
                                        const float*, int, const gfo_frame_bounds*, const gfo_map_point*,
                                        const uint8_t*, int, float, float, const uint8_t*, int32_t*, int32_t*, int*)
{
    if (c) c->err = "gfo_search_by_projection: device path not built yet";
    return GFO_ERR_STATE;
}
