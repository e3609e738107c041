/* lmx_types.h — plain-C POD layouts shared by the C ABI (lumix_mi355.h), the CPU oracle (oracle/) and tests.
 *
 * Every struct here is byte-compatible with the LumixEngine type it names, so an engine-side adapter can
 * reinterpret_cast instead of converting. Citations are relative to the reference tree (src/...).
 */
#ifndef LMX_TYPES_H
#define LMX_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frustum plane slots, core/geometry.h:73-83 (Frustum::Planes). EXTRA0/EXTRA1 duplicate NEAR
 * (core/geometry.cpp:134-136, 343-345). */
enum {
	LMX_PLANE_NEAR = 0,
	LMX_PLANE_FAR = 1,
	LMX_PLANE_LEFT = 2,
	LMX_PLANE_RIGHT = 3,
	LMX_PLANE_TOP = 4,
	LMX_PLANE_BOTTOM = 5,
	LMX_PLANE_EXTRA0 = 6,
	LMX_PLANE_EXTRA1 = 7,
	LMX_PLANE_COUNT = 8
};

/* RenderableTypes, renderer/render_module.h:293-301. The u8 `type` of a culling cell / CullResult page. */
enum {
	LMX_TYPE_MESH = 0,
	LMX_TYPE_DECAL = 1,
	LMX_TYPE_LOCAL_LIGHT = 2,
	LMX_TYPE_CURVE_DECAL = 3,
	LMX_TYPE_PARTICLES = 4,
	LMX_TYPE_COUNT = 5,
	LMX_TYPE_ALL = 0xff /* culling_system.cpp:310-319: 0xff is reserved for "all types" */
};

/* ShiftedFrustum, core/geometry.h:102-153: 8 planes SoA + 8 corner points (fp32, relative to `origin`) + fp64 origin.
 * sizeof == 256, alignas(16). */
typedef struct LmxShiftedFrustum {
	float xs[LMX_PLANE_COUNT];
	float ys[LMX_PLANE_COUNT];
	float zs[LMX_PLANE_COUNT];
	float ds[LMX_PLANE_COUNT];
	float points[8][3];
	double origin[3];
	double _pad; /* alignas(16) tail padding of the reference struct */
} LmxShiftedFrustum;

/* Frustum, core/geometry.h:29-99 (the cell-relative result of ShiftedFrustum::getRelative). sizeof == 224. */
typedef struct LmxFrustum {
	float xs[LMX_PLANE_COUNT];
	float ys[LMX_PLANE_COUNT];
	float zs[LMX_PLANE_COUNT];
	float ds[LMX_PLANE_COUNT];
	float points[8][3];
} LmxFrustum;

/* Transform, core/math.h:306-327: fp64 position, fp32 quaternion (x,y,z,w), fp32 non-uniform scale. sizeof == 56. */
typedef struct LmxTransform {
	double pos[3];
	float rot[4];
	float scale[3];
	float _pad;
} LmxTransform;

/* LocalRigidTransform, core/math.h:262-270: fp32 position + quaternion. sizeof == 28. */
typedef struct LmxLocalRigidTransform {
	float pos[3];
	float rot[4];
} LmxLocalRigidTransform;

/* Matrix, core/math.h:329-393: column-major 4x4 fp32, columns[c] = {x,y,z,w}. sizeof == 64. */
typedef struct LmxMatrix {
	float columns[4][4];
} LmxMatrix;

/* Mesh::Skin, renderer/model.h:81-84. sizeof == 24. */
typedef struct LmxSkin {
	float weights[4];
	int16_t indices[4];
} LmxSkin;

/* Viewport, core/geometry.h:177-199 (only the fields Viewport::getFrustum() reads). Not layout-compatible. */
typedef struct LmxViewport {
	int32_t is_ortho;
	float fov;
	float ortho_size;
	int32_t w;
	int32_t h;
	double pos[3];
	float rot[4];
	float near_plane;
	float far_plane;
} LmxViewport;

enum {
	LMX_CULL_CELL_SIZE = 300,        /* culling_system.cpp:75 */
	LMX_CULL_PAGE_SPHERES = 201,     /* CellPage::MAX_COUNT, culling_system.cpp:59 */
	LMX_CULLRESULT_PAGE_IDS = 1020,  /* culling_system.h:55 */
	LMX_MAX_BONES = 196              /* Model::Bone::MAX_COUNT, renderer/model.h:155 */
};

/* ---- createSortKeys inputs (renderer/pipeline.cpp:3789-3968) ---- */

/* LODMeshIndices, renderer/model.h:129-133 */
typedef struct LmxLodIndices {
	int32_t from, to;
} LmxLodIndices;

/* The fields of Model that createSortKeys reads: m_lod_distances[MAX_LOD_COUNT] (squared distances, model.h:234),
 * m_lod_indices[MAX_LOD_COUNT + 1] (model.h:233; entry 4 stays {0, -1}, model.cpp:98) and Mesh::type of its meshes. */
typedef struct LmxKeysModel {
	float lod_distances[4];
	LmxLodIndices lod_indices[5];
	uint32_t first_mesh; /* into the mesh-type table of lmx_keys_set_models */
	uint32_t mesh_count;
} LmxKeysModel;

/* MeshMaterial (renderer/model.h:59-68) as createSortKeys sees it: sort_key and material->getLayer(). */
typedef struct LmxMeshMaterial {
	uint32_t sort_key;
	uint8_t layer;
	uint8_t _pad[3];
} LmxMeshMaterial;

enum { LMX_MESH_RIGID = 0, LMX_MESH_SKINNED = 1 };         /* Mesh::Type, renderer/model.h:86-89 */
enum { LMX_MODEL_INSTANCE_MOVED = 1 << 3 };                  /* ModelInstance::MOVED, renderer/render_module.h:212 */

/* DrawCommandTypes, renderer/pipeline.cpp:41-51 (bits 32..36 of a sort value) */
enum {
	LMX_DRAW_MESH = 0,
	LMX_DRAW_AUTOINSTANCED = 1,
	LMX_DRAW_SKINNED = 2,
	LMX_DRAW_DECAL = 3,
	LMX_DRAW_CURVE_DECAL = 4
};
#define LMX_SORT_KEY_BUCKET_SHIFT 56                         /* pipeline.cpp:70 */
#define LMX_SORT_KEY_INSTANCED_FLAG (1ull << 55)             /* pipeline.cpp:71 */
#define LMX_SORT_VALUE_INSTANCER_SHIFT 16                    /* pipeline.cpp:73 */
#define LMX_SORT_VALUE_MESH_IDX_SHIFT 40                     /* pipeline.cpp:76 */
#define LMX_SORT_VALUE_TYPE_SHIFT 32                         /* pipeline.cpp:77 */

/* The per-view state createSortKeys reads (pipeline.cpp:3797-3832). */
typedef struct LmxKeysView {
	double camera_pos[3];             /* view.cp.pos */
	double lod_ref_point[3];          /* m_viewport.pos */
	float lod_multiplier;             /* Renderer::getLODMultiplier() */
	float time_delta;                 /* Engine::getLastTimeDelta() */
	uint32_t frame_number;            /* Renderer::frameNumber() % 0xffFFffFF */
	uint8_t is_shadow;                /* view.cp.is_shadow */
	uint8_t layer_to_bucket[255];     /* View::layer_to_bucket, 0xff = no bucket renders the layer (pipeline.cpp:1003-1020) */
	uint8_t bucket_depth_sorted[256]; /* buckets[b].sort == BucketDesc::DEPTH */
} LmxKeysView;

/* ---- animation sampling inputs (animation/animation.h:86-115, animation.cpp:29-204) ---- */

typedef struct LmxAnimConstTranslation { /* Animation::ConstTranslationTrack */
	float value[3];
	uint16_t bone_index;
	uint16_t _pad;
} LmxAnimConstTranslation;

typedef struct LmxAnimTranslationTrack { /* Animation::TranslationTrack: bit-packed, value = min + to_range * bits (in fp64, animation.cpp:313-316) */
	float min[3];
	float to_range[3];
	uint16_t offset_bits;
	uint16_t bone_index;
	uint8_t bitsizes[3];
	uint8_t _pad;
} LmxAnimTranslationTrack;

typedef struct LmxAnimConstRotation { /* Animation::ConstRotationTrack */
	float value[4];
	uint16_t bone_index;
	uint16_t _pad;
} LmxAnimConstRotation;

typedef struct LmxAnimRotationTrack { /* Animation::RotationTrack: 3 packed channels + sign bit, the skipped one rebuilt from the norm */
	float min[3];
	float to_range[3];
	uint16_t offset_bits;
	uint16_t bone_index;
	uint8_t bitsizes[3];
	uint8_t skipped_channel;
} LmxAnimRotationTrack;

/* What AnimationSampler reads of an Animation resource. Streams hold frame_count + 1 frames (animation.cpp:464). */
typedef struct LmxAnimation {
	float fps;                               /* m_fps */
	uint32_t frame_count;                    /* m_frame_count */
	uint32_t length;                         /* getLength().raw(): Time units of 1 / 32768 s (animation.h:41) */
	uint32_t translations_frame_size_bits;   /* m_translations_frame_size_bits */
	uint32_t rotations_frame_size_bits;      /* m_rotations_frame_size_bits */
	uint32_t n_const_translations, n_translations, n_const_rotations, n_rotations;
	const LmxAnimConstTranslation* const_translations;
	const LmxAnimTranslationTrack* translations;
	const LmxAnimConstRotation* const_rotations;
	const LmxAnimRotationTrack* rotations;
	const uint8_t* translation_stream;
	uint64_t translation_stream_size;
	const uint8_t* rotation_stream;
	uint64_t rotation_stream_size;
	int32_t root_translation_track;          /* RootMotion::translation_track_idx, -1 = none (animation.cpp:320) */
	int32_t root_rotation_track;             /* RootMotion::rotation_track_idx, -1 = none (animation.cpp:33) */
	const float* root_pose_translations;     /* RootMotion::pose_translations, (frame_count + 1) x 3 */
	const float* root_pose_rotations;        /* RootMotion::pose_rotations, (frame_count + 1) x 4 */
} LmxAnimation;

/* One SAMPLE instruction of an Animator's blend stack (anim::BlendStackInstructions::SAMPLE, controller.cpp:282-289: slot, weight,
 * time, looped as the controller's nodes wrote them; `animation` is the id lmx_anim_add returned for RuntimeContext::animations[slot]). */
typedef struct LmxBlendSample {
	uint32_t animation;
	float weight;
	uint32_t time;                           /* Time units; wrapped (looped) or clamped to the animation's length by getPose, controller.cpp:148 */
	uint32_t looped;
} LmxBlendSample;

#define LMX_TIME_ONE_SECOND (1u << 15)       /* Time::ONE_SECOND, animation/animation.h:41 */
#define LMX_ANIM_NONE 0xffffffffu

#ifdef __cplusplus
}
#endif

#endif
