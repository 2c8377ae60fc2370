// VaeModel arrives with vae.cpp
#include "ctx.h"
namespace star { struct VaeModel { int unused = 0; }; }
