// Forwarding header: lets reference-style `#include "sphere.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
