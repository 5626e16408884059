vec2 = vec3 = uvec3 = None
