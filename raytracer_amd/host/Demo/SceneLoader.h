// helpers::LoadScene -- the JSON scene ingestion of the reference's Demo (Demo/SceneLoader.h:5-8, SceneLoader.cpp).
#pragma once

#include "../Core/Scene/Scene.h"
#include "../Core/Scene/Camera.h"

#include <string>

namespace helpers {

RAYLIB_API bool LoadScene(const std::string& path, rt::Scene& scene, rt::Camera& camera);

} // namespace helpers
