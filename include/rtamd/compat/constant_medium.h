// Forwarding header: lets reference-style `#include "constant_medium.h"` resolve to the MI355X scene API.
#include "../scene_api.h"
