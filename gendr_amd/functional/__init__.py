from .renderer import render, soft_rasterize, GenDRFunction
from .geometry import (look_at, look, perspective, orthogonal, face_vertices, vertex_normals,
                       get_points_from_angles)
from .lighting import ambient_lighting, directional_lighting
from .obj_io import load_obj, save_obj, save_voxel, voxelization
from .projection import CameraFacesFunction, ProjectFacesFunction, project_faces, look_at_faces, look_faces
from .shading import LightFacesFunction, light_faces, light_params
from .silhouette import render_silhouette, silhouette_iou, silhouette_iou_loss, SilhouetteFunction, SilhouetteIoUFunction
