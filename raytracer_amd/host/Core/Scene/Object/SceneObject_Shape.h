// reference include path compatibility
#pragma once
#include "SceneObject.h"
