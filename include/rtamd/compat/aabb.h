// Forwarding header: lets reference-style `#include "aabb.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
