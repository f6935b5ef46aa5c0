// Forwarding header: lets reference-style `#include "stb_image_utils.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
