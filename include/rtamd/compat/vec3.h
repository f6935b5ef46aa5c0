// Forwarding header: lets reference-style `#include "vec3.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
