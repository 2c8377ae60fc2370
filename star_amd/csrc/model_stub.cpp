// temporary: UNetModel / VaeModel definitions arrive with unet.cpp / vae.cpp
#include "ctx.h"
namespace star { struct UNetModel {}; struct VaeModel {}; }
