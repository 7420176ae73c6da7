// Drop-in for the reference header of the same name: forwards to the device-backed mirror.
#include "ray_amd_host.h"
