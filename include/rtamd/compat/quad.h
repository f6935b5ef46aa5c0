// Forwarding header: lets reference-style `#include "quad.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
