// reference include path compatibility: Core/Scene/Light/DirectionalLight.h
#pragma once
#include "Light.h"
