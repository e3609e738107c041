#!/usr/bin/env python
"""TEST INFRASTRUCTURE. Cuts PipelineImpl::createSortKeys and the records it works on out of the reference tree, where it lies, into
include fragments under a temporary directory (oracle/Makefile deletes it after the compile; nothing is committed), so that
oracle/ref/keys_shim.cpp can compile the REFERENCE'S OWN key-building code into oracle/_ref/liblmx_ref.so:

    keys_consts.inc          src/renderer/pipeline.cpp     DrawCommandTypes, floatFlip, SORT_KEY_* / SORT_VALUE_*, make*SortKey / make*SortValue
    keys_sorter.inc          src/renderer/pipeline.cpp     PipelineImpl::Sorter (+ Inserter)
    keys_instancer.inc       src/renderer/pipeline.cpp     PipelineImpl::AutoInstancer
    keys_view.inc            src/renderer/pipeline.cpp     PipelineImpl::View
    keys_head.inc            src/renderer/pipeline.cpp     createSortKeys: everything before jobs::runOnWorkers (instancers, bucket_map, frame number)
    keys_worker.inc          src/renderer/pipeline.cpp     createSortKeys: the worker lambda's body up to (not including) "fill instance data"
    keys_cull_result.inc     src/renderer/culling_system.h struct CullResult
    keys_paged_iter.inc      src/core/page_allocator.h     PagedListIterator
    keys_model_structs.inc   src/renderer/model.h          MaterialIndex, MeshMaterial, LODMeshIndices
    keys_mesh_type.inc       src/renderer/model.h          Mesh::Type
    keys_lod_fn.inc          src/renderer/model.h          Model::getLODMeshIndices
    keys_model_instance.inc  src/renderer/render_module.h  ModelInstance, RenderableTypes
    keys_bucket_sort.inc     src/renderer/pipeline.h       BucketDesc::Sort
    keys_camera_params.inc   src/renderer/pipeline.h       CameraParams

pipeline.cpp cannot be compiled whole: it is the renderer (gpu back end, resources, draw streams, job system). Pieces are located by
anchor strings + brace matching; the script fails loudly if an anchor is missing.

    python oracle/ref/slice_sort_keys.py /root/reference <tmp>/gen
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from slice_animation import block  # noqa: E402


def between(text, start, end, include_start=True):
    a = text.index(start)
    b = text.index(end, a)
    return text[a if include_start else a + len(start):b]


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    src = os.path.join(ref, "src")

    def put(name, text):
        open(os.path.join(out, name), "w").write(text + "\n")

    pc = open(os.path.join(src, "renderer", "pipeline.cpp")).read()
    last_maker = block(pc, "LUMIX_FORCE_INLINE SortValue makeAutoInstancedSortValue(")
    consts = pc[pc.index("enum class DrawCommandTypes : u8 {"):pc.index(last_maker) + len(last_maker)]
    assert "floatFlip" in consts and "SORT_VALUE_MESH_IDX_SHIFT" in consts and "makeDepthSortKey" in consts
    put("keys_consts.inc", consts)
    put("keys_sorter.inc", block(pc, "struct Sorter {", trailer=";"))
    put("keys_instancer.inc", block(pc, "struct AutoInstancer {", trailer=";"))
    put("keys_view.inc", block(pc, "struct View {", trailer=";"))
    fn = block(pc, "void createSortKeys(PipelineImpl::View& view) {")
    head = between(fn, "PagedListIterator<const CullResult> iterator(view.renderables);", "jobs::runOnWorkers([&](){")
    assert "bucket_map[i] |= 0x100" in head and "frame_number" in head
    put("keys_head.inc", head)
    worker = between(fn, 'PROFILE_BLOCK("create keys");', 'PROFILE_BLOCK("fill instance data");', include_start=False)
    assert "makeAutoInstancedSortKey" in worker and "getLODMeshIndices" in worker and "compareExchange" in worker
    put("keys_worker.inc", worker)

    ch = open(os.path.join(src, "renderer", "culling_system.h")).read()
    put("keys_cull_result.inc", block(ch, "struct CullResult {", trailer=";"))
    ph = open(os.path.join(src, "core", "page_allocator.h")).read()
    it = block(ph, "struct PagedListIterator", trailer=";")
    put("keys_paged_iter.inc", "template <typename T>\n" + it)

    mh = open(os.path.join(src, "renderer", "model.h")).read()
    idx_line = "enum class MaterialIndex : u32 {};"
    assert idx_line in mh
    put("keys_model_structs.inc", idx_line + "\n\n" + block(mh, "struct MeshMaterial {", trailer=";") + "\n\n" + block(mh, "struct LODMeshIndices", trailer=";"))
    mesh = block(mh, "struct LUMIX_RENDERER_API Mesh {", trailer=";")
    put("keys_mesh_type.inc", block(mesh, "enum Type : u8 {", trailer=";"))
    put("keys_lod_fn.inc", block(mh, "u32 getLODMeshIndices(float squared_distance) const {"))

    rh = open(os.path.join(src, "renderer", "render_module.h")).read()
    put("keys_model_instance.inc", block(rh, "enum class RenderableTypes : u8 {", trailer=";") + "\n\n" + block(rh, "struct ModelInstance {", trailer=";"))

    plh = open(os.path.join(src, "renderer", "pipeline.h")).read()
    desc = block(plh, "struct BucketDesc {", trailer=";")
    put("keys_bucket_sort.inc", block(desc, "enum Sort {", trailer=";"))
    put("keys_camera_params.inc", block(plh, "struct CameraParams {", trailer=";"))
    print("sliced", sorted(f for f in os.listdir(out) if f.startswith("keys_")))


if __name__ == "__main__":
    main()
