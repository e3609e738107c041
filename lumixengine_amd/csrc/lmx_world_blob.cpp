// lmx_world_blob.cpp — host-side ingest of a serialized World (engine/world.cpp:837-1043) into the inputs of lmx_world_build:
// header, module list, flags, the LZ4 block (Engine::compress = LZ4 block format, engine.cpp:254-269), entity transforms,
// entity names (skipped), hierarchy records. Module payloads that follow are not read. Pure host code, no device needed.
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lumix_mi355.h"

namespace {

constexpr uint32_t WORLD_MAGIC = 0x4c57524cu; // 'LWRL' as an MSVC/GCC multi-character constant, world.cpp:830
constexpr uint32_t WORLD_VERSION_LATEST = 6;  // WorldVersion::LATEST, world.h:17-26 (> COMPRESSED = 5)
constexpr uint32_t WORLD_HAS_PARTITIONS = 1;  // WorldSerializeFlags::HAS_PARTITIONS

struct Reader {
	const uint8_t* p;
	size_t size, pos = 0;
	bool overflow = false;
	template <typename T> T read() {
		T v{};
		if (pos + sizeof(T) > size) { overflow = true; pos = size; return v; }
		memcpy(&v, p + pos, sizeof(T));
		pos += sizeof(T);
		return v;
	}
	void skip(size_t n) { if (pos + n > size) { overflow = true; pos = size; } else pos += n; }
	bool skip_string() { // InputMemoryStream::readString, core/stream.cpp:424-436
		while (pos < size && p[pos]) ++pos;
		if (pos >= size) { overflow = true; return false; }
		++pos;
		return true;
	}
};

// LZ4 block format (what LZ4_decompress_safe accepts): sequences of {token, [literal length bytes], literals, 16-bit little-endian
// offset, [match length bytes]}; the last sequence ends after its literals. Returns the number of bytes written, or -1.
long lz4_block_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap) {
	size_t ip = 0, op = 0;
	while (ip < src_size) {
		const uint8_t token = src[ip++];
		size_t lit = token >> 4;
		if (lit == 15) {
			uint8_t b;
			do {
				if (ip >= src_size) return -1;
				b = src[ip++];
				lit += b;
			} while (b == 255);
		}
		if (ip + lit > src_size || op + lit > dst_cap) return -1;
		memcpy(dst + op, src + ip, lit);
		ip += lit;
		op += lit;
		if (ip >= src_size) break; // last sequence: literals only
		if (ip + 2 > src_size) return -1;
		const size_t offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
		ip += 2;
		if (offset == 0 || offset > op) return -1;
		size_t len = token & 15;
		if (len == 15) {
			uint8_t b;
			do {
				if (ip >= src_size) return -1;
				b = src[ip++];
				len += b;
			} while (b == 255);
		}
		len += 4; // MINMATCH
		if (op + len > dst_cap) return -1;
		for (size_t i = 0; i < len; ++i) dst[op + i] = dst[op + i - offset]; // byte-wise: matches may overlap their own output
		op += len;
	}
	return (long)op;
}

struct Parsed {
	LmxWorldBlobInfo info;
	std::vector<uint8_t> blob;
	std::vector<std::string> module_names; // serializeModuleList, world.cpp:780-786: what the payload section may contain, in any order
	size_t entities_at = 0, hierarchy_at = 0;
	bool partitions = false;
};

int parse(const void* data, size_t size, Parsed& out) {
	memset(&out.info, 0, sizeof(out.info));
	if (!data) return LMX_ERR_INVALID_ARGUMENT;
	Reader in{(const uint8_t*)data, size};
	const uint32_t magic = in.read<uint32_t>();
	const uint32_t version = in.read<uint32_t>();
	if (in.overflow || magic != WORLD_MAGIC) return LMX_ERR_INVALID_ARGUMENT; // "Wrong or corrupted file" (legacy headers are not supported)
	if (version != WORLD_VERSION_LATEST) return LMX_ERR_INVALID_ARGUMENT;    // only the current, compressed layout
	out.info.version = version;
	const int32_t n_modules = in.read<int32_t>(); // serializeModuleList, world.cpp:780-786
	if (n_modules < 0) return LMX_ERR_INVALID_ARGUMENT;
	out.module_names.clear();
	for (int32_t i = 0; i < n_modules; ++i) {
		const size_t from = in.pos;
		if (!in.skip_string()) return LMX_ERR_INVALID_ARGUMENT;
		out.module_names.emplace_back(reinterpret_cast<const char*>(in.p + from));
	}
	out.info.n_modules = (uint32_t)n_modules;
	out.info.flags = in.read<uint32_t>();
	out.partitions = (out.info.flags & WORLD_HAS_PARTITIONS) != 0;
	out.info.uncompressed_size = in.read<uint32_t>();
	out.info.compressed_size = in.read<uint32_t>();
	if (in.overflow || in.pos + out.info.compressed_size > size) return LMX_ERR_INVALID_ARGUMENT;
	// an LZ4 block expands at most 255 x (a run of 0xff length bytes per 255 literals / match bytes): a header that claims more is
	// corrupt, and the claim must not be handed to the allocator unchecked (up to 4 GiB from an untrusted file)
	if ((uint64_t)out.info.uncompressed_size > (uint64_t)out.info.compressed_size * 255u + 64u) return LMX_ERR_INVALID_ARGUMENT;
	try {
		out.blob.resize(out.info.uncompressed_size);
	} catch (const std::bad_alloc&) {
		return LMX_ERR_OUT_OF_MEMORY;
	}
	const long got = lz4_block_decode(in.p + in.pos, out.info.compressed_size, out.blob.data(), out.blob.size());
	if (got != (long)out.blob.size()) return LMX_ERR_INVALID_ARGUMENT; // Engine::decompress: result == output.length()

	Reader s{out.blob.data(), out.blob.size()};
	(void)s.read<uint32_t>(); // entity_map.reserve(to_reserve)
	out.entities_at = s.pos;
	uint32_t n_entities = 0, max_index = 0;
	for (;;) { // world.cpp:956-973
		const int32_t e = s.read<int32_t>();
		if (s.overflow) return LMX_ERR_INVALID_ARGUMENT;
		if (e < 0) break; // INVALID_ENTITY
		s.skip(24 + 16 + 12 + (out.partitions ? 2 : 0)); // DVec3 pos, Quat rot, Vec3 scale [, PartitionHandle u16]
		++n_entities;
		if ((uint32_t)e > max_index) max_index = (uint32_t)e;
	}
	out.info.n_entities = n_entities;
	out.info.max_entity_index = max_index;
	const uint32_t n_names = s.read<uint32_t>(); // :975-983
	for (uint32_t i = 0; i < n_names && !s.overflow; ++i) {
		s.skip(4);
		s.skip_string();
	}
	out.info.n_names = n_names;
	out.info.n_hierarchy = s.read<uint32_t>(); // :985-1013
	out.hierarchy_at = s.pos;
	s.skip((size_t)out.info.n_hierarchy * (16 + 24 + 16 + 12));
	if (s.overflow) return LMX_ERR_INVALID_ARGUMENT; // "End of file encountered while trying to read data"
	return LMX_OK;
}

// ---- the renderer module's payload (render_module.cpp:1225-1250) ------------------------------------------------------------
constexpr int32_t RENDER_VERSION_MIN = 16, RENDER_VERSION_LATEST = 18; // RenderModuleVersion, render_module.h:303-324

struct RenderParsed {
	LmxRenderBlobInfo info;
	size_t paths_at = 0, instances_at = 0, attachments_at = 0;
};

// true when `at` is where a module payload may end: the end of the blob or the header of another module of the file's module list
// (its name as a NUL-terminated string followed by a plausible i32 version)
bool at_module_boundary(const Parsed& p, size_t at) {
	if (at == p.blob.size()) return true;
	for (const std::string& n : p.module_names) {
		const size_t len = n.size();
		if (at + len + 1 + 4 > p.blob.size() || memcmp(p.blob.data() + at, n.c_str(), len + 1) != 0) continue;
		int32_t v;
		memcpy(&v, p.blob.data() + at + len + 1, 4);
		if (v >= 0 && v <= 255) return true;
	}
	return false;
}

// Module payloads carry no size: a module is found by its header - the name as a NUL-terminated string (preceded by the i32
// module count or by the previous payload) followed by a plausible i32 version. A byte sequence inside an earlier payload can look
// like one: the scan only accepts names the file's own module list holds, and the renderer walk below re-checks that the payload it
// walked ENDS on a module boundary (a false start would have to end on one too).
bool find_module(const Parsed& p, const char* name, size_t* payload_at, int32_t* version) {
	bool listed = false;
	for (const std::string& n : p.module_names) listed |= n == name;
	if (!listed) return false;
	Reader s{p.blob.data(), p.blob.size(), p.hierarchy_at};
	s.skip((size_t)p.info.n_hierarchy * (16 + 24 + 16 + 12));
	const int32_t n_modules = s.read<int32_t>();
	if (s.overflow || n_modules <= 0) return false;
	const size_t len = strlen(name), first = s.pos;
	for (size_t at = first; at + len + 1 + 4 <= p.blob.size(); ++at) {
		if (memcmp(p.blob.data() + at, name, len + 1) != 0) continue;
		if (at != first && at < first + 5) continue; // (a later module's name starts behind the previous module's version + payload)
		int32_t v;
		memcpy(&v, p.blob.data() + at + len + 1, 4);
		if (v < 0 || v > 255) continue;
		*payload_at = at + len + 1 + 4;
		*version = v;
		return true;
	}
	return false;
}

int parse_renderer(const void* data, size_t size, Parsed& p, RenderParsed& out) {
	memset(&out.info, 0, sizeof(out.info));
	if (int rc = parse(data, size, p)) return rc;
	size_t at = 0;
	int32_t version = 0;
	if (!find_module(p, "renderer", &at, &version)) return LMX_ERR_INVALID_ARGUMENT;
	LmxRenderBlobInfo& info = out.info;
	info.version = version;
	info.payload_offset = (uint32_t)at;
	if (version < RENDER_VERSION_MIN || version > RENDER_VERSION_LATEST) return LMX_ERR_INVALID_ARGUMENT; // older layouts are not restated
	Reader s{p.blob.data(), p.blob.size(), at};
	// cameras (:979-1014): entity, fov, near, far, ortho_size, screen_width, screen_height, is_ortho, film grain, 5 depth-of-field fields
	info.n_cameras = s.read<uint32_t>();
	for (uint32_t i = 0; i < info.n_cameras && !s.overflow; ++i) s.skip(4 + 6 * 4 + 1 + 4 + 1 + 4 * 4);
	// model instances (:1051-1098): path table, then per entity slot flags (u8) [+ path offset + material overrides when VALID]
	info.model_paths_size = s.read<uint32_t>();
	out.paths_at = s.pos;
	s.skip(info.model_paths_size);
	info.n_model_instance_slots = s.read<uint32_t>();
	out.instances_at = s.pos;
	for (uint32_t i = 0; i < info.n_model_instance_slots && !s.overflow; ++i) {
		const uint8_t flags = s.read<uint8_t>();
		if (!(flags & 4u)) continue; // ModelInstance::VALID
		++info.n_model_instances;
		s.skip(4);
		if (version > 15) { // MATERIAL_OVERRIDE: one path per mesh
			const uint32_t n = s.read<uint32_t>();
			for (uint32_t k = 0; k < n && !s.overflow; ++k) s.skip_string();
		} else {
			s.skip_string();
		}
	}
	// lights (:1100-1161): PointLight raw (48 B), environments field by field, the active global light
	info.n_point_lights = s.read<uint32_t>();
	s.skip((size_t)info.n_point_lights * 48);
	info.n_environments = s.read<uint32_t>();
	for (uint32_t i = 0; i < info.n_environments && !s.overflow; ++i) {
		s.skip(12 + 4 + 4 + 4 + 16 + 4); // light_color, direct, indirect, entity, cascades, flags
		s.skip_string();                 // sky cubemap
		s.skip(4 + 5 * 12 + 6 * 4 + 1);  // sky intensity, 5 colours, sunlight strength .. fog_top, atmo_enabled
		s.skip(1);                       // godrays_enabled
		s.skip(1 + 4 + 4);               // clouds (> CLOUDS)
		s.skip(4);                       // fog_density (> FOG_DENSITY)
	}
	s.skip(4);
	// terrains (:1213-1224, terrain.cpp:323-357)
	info.n_terrains = (uint32_t)s.read<int32_t>();
	for (uint32_t i = 0; i < info.n_terrains && !s.overflow; ++i) {
		s.skip(4 + 8);
		s.skip_string();
		s.skip(4 + 4 + 4 + 4);
		const int32_t grass = s.read<int32_t>();
		for (int32_t k = 0; k < grass && !s.overflow; ++k) {
			s.skip_string();
			s.skip(4 + 4 + 4);
		}
	}
	// particle systems (:919-934, particle_system.cpp:463-475): entity, autodestroy, resource path
	info.n_particle_systems = s.read<uint32_t>();
	for (uint32_t i = 0; i < info.n_particle_systems && !s.overflow; ++i) {
		s.skip(4 + 1);
		s.skip_string();
	}
	// bone attachments (:895-914)
	info.n_bone_attachments = s.read<uint32_t>();
	out.attachments_at = s.pos;
	s.skip((size_t)info.n_bone_attachments * ((version > 17 ? 8 : 4) + 4 + 4 + 28));
	// probes (:877-892 raw EnvironmentProbe 136 B; :829-847 guid, flags, size, half_extents), decals (:731-775)
	info.n_environment_probes = s.read<uint32_t>();
	s.skip((size_t)info.n_environment_probes * (4 + 136));
	info.n_reflection_probes = s.read<uint32_t>();
	s.skip((size_t)info.n_reflection_probes * (4 + 8 + 4 + 4 + 12));
	info.n_decals = s.read<uint32_t>();
	for (uint32_t i = 0; i < info.n_decals && !s.overflow; ++i) {
		s.skip(4 + 12 + 8);
		s.skip_string();
	}
	info.n_curve_decals = s.read<uint32_t>();
	for (uint32_t i = 0; i < info.n_curve_decals && !s.overflow; ++i) {
		s.skip(4 + 8 + 4 + 8 + 8);
		s.skip_string();
	}
	if (version <= 16 && s.read<uint32_t>() != 0) return LMX_ERR_INVALID_ARGUMENT; // deserializeFurs (:725-729): the count must be 0
	info.n_instanced_models = s.read<uint32_t>(); // (:703-723) entity, model path, instances (32 B each)
	for (uint32_t i = 0; i < info.n_instanced_models && !s.overflow; ++i) {
		s.skip(4);
		s.skip_string();
		const uint32_t n = s.read<uint32_t>();
		s.skip((size_t)n * 32);
	}
	info.n_procedural_geometries = s.read<uint32_t>();
	if (s.overflow) return LMX_ERR_INVALID_ARGUMENT;
	info.payload_size = info.n_procedural_geometries ? 0u : (uint32_t)(s.pos - at); // vertex declarations are not walked
	// the walk must end where the next module begins (or the blob ends): a payload found at a look-alike byte sequence does not
	if (!info.n_procedural_geometries && !at_module_boundary(p, s.pos)) return LMX_ERR_INVALID_ARGUMENT;
	return LMX_OK;
}

} // namespace

extern "C" {

int lmx_world_blob_find_module(const void* data, size_t size, const char* name, uint32_t* payload_offset, int32_t* version) {
	if (!name || !name[0]) return LMX_ERR_INVALID_ARGUMENT;
	Parsed p;
	if (int rc = parse(data, size, p)) return rc;
	size_t at = 0;
	int32_t v = 0;
	if (!find_module(p, name, &at, &v)) return LMX_ERR_INVALID_ARGUMENT;
	if (payload_offset) *payload_offset = (uint32_t)at;
	if (version) *version = v;
	return LMX_OK;
}

int lmx_render_blob_info(const void* data, size_t size, LmxRenderBlobInfo* out) {
	if (!out) return LMX_ERR_INVALID_ARGUMENT;
	Parsed p;
	RenderParsed r;
	const int rc = parse_renderer(data, size, p, r);
	*out = r.info;
	return rc;
}

int lmx_render_blob_read_bone_attachments(const void* data, size_t size, uint32_t cap, LmxBlobBoneAttachment* out) {
	Parsed p;
	RenderParsed r;
	if (int rc = parse_renderer(data, size, p, r)) return rc;
	if (r.info.n_bone_attachments > cap) return LMX_ERR_CAPACITY;
	if (r.info.n_bone_attachments && !out) return LMX_ERR_INVALID_ARGUMENT;
	Reader s{p.blob.data(), p.blob.size(), r.attachments_at};
	for (uint32_t i = 0; i < r.info.n_bone_attachments; ++i) {
		LmxBlobBoneAttachment a;
		memset(&a, 0, sizeof(a));
		a.bone_name_hash = r.info.version > 17 ? s.read<uint64_t>() : (uint64_t)(uint32_t)s.read<int32_t>();
		a.entity = s.read<int32_t>();
		a.parent_entity = s.read<int32_t>();
		for (int k = 0; k < 3; ++k) a.relative.pos[k] = s.read<float>();
		for (int k = 0; k < 4; ++k) a.relative.rot[k] = s.read<float>();
		out[i] = a;
	}
	return s.overflow ? LMX_ERR_INVALID_ARGUMENT : LMX_OK;
}

int lmx_render_blob_read_model_instances(const void* data, size_t size, uint32_t n_slots, uint8_t* flags, uint32_t* path_offset, char* paths, uint32_t paths_cap) {
	if (!flags || !path_offset) return LMX_ERR_INVALID_ARGUMENT;
	Parsed p;
	RenderParsed r;
	if (int rc = parse_renderer(data, size, p, r)) return rc;
	if (n_slots < r.info.n_model_instance_slots || (r.info.model_paths_size && (!paths || paths_cap < r.info.model_paths_size))) return LMX_ERR_CAPACITY;
	// callers strlen() paths + path_offset[i]: the table must end in a NUL (every offset below is checked to lie inside it)
	if (r.info.model_paths_size && p.blob[r.paths_at + r.info.model_paths_size - 1] != 0) return LMX_ERR_INVALID_ARGUMENT;
	if (r.info.model_paths_size) memcpy(paths, p.blob.data() + r.paths_at, r.info.model_paths_size);
	for (uint32_t e = 0; e < n_slots; ++e) {
		flags[e] = 0;
		path_offset[e] = 0xffffffffu;
	}
	Reader s{p.blob.data(), p.blob.size(), r.instances_at};
	for (uint32_t e = 0; e < r.info.n_model_instance_slots; ++e) {
		const uint8_t f = s.read<uint8_t>();
		if (!(f & 4u)) continue;
		flags[e] = f;
		path_offset[e] = s.read<uint32_t>();
		if (path_offset[e] != 0xffffffffu && path_offset[e] >= r.info.model_paths_size) return LMX_ERR_INVALID_ARGUMENT;
		if (r.info.version > 15) {
			const uint32_t n = s.read<uint32_t>();
			for (uint32_t k = 0; k < n && !s.overflow; ++k) s.skip_string();
		} else {
			s.skip_string();
		}
	}
	return s.overflow ? LMX_ERR_INVALID_ARGUMENT : LMX_OK;
}

int lmx_world_blob_info(const void* data, size_t size, LmxWorldBlobInfo* out) {
	if (!out) return LMX_ERR_INVALID_ARGUMENT;
	Parsed p;
	const int rc = parse(data, size, p);
	*out = p.info;
	return rc;
}

int lmx_world_blob_read(const void* data, size_t size, uint32_t n_slots, int32_t* parent, LmxTransform* transforms, LmxTransform* world, uint8_t* valid) {
	if (!parent || !transforms) return LMX_ERR_INVALID_ARGUMENT;
	Parsed p;
	if (int rc = parse(data, size, p)) return rc;
	if (p.info.n_entities && n_slots <= p.info.max_entity_index) return LMX_ERR_CAPACITY;
	for (uint32_t e = 0; e < n_slots; ++e) { // slots the file does not mention: detached identity entities, valid = 0
		parent[e] = -1;
		memset(&transforms[e], 0, sizeof(LmxTransform));
		transforms[e].rot[3] = 1.f;
		transforms[e].scale[0] = transforms[e].scale[1] = transforms[e].scale[2] = 1.f;
		if (world) world[e] = transforms[e];
		if (valid) valid[e] = 0;
	}
	Reader s{p.blob.data(), p.blob.size(), p.entities_at};
	auto read_transform = [&](LmxTransform* t) {
		memset(t, 0, sizeof(*t));
		for (int k = 0; k < 3; ++k) t->pos[k] = s.read<double>();
		for (int k = 0; k < 4; ++k) t->rot[k] = s.read<float>();
		for (int k = 0; k < 3; ++k) t->scale[k] = s.read<float>();
	};
	for (uint32_t i = 0; i < p.info.n_entities; ++i) {
		const int32_t e = s.read<int32_t>();
		LmxTransform t;
		read_transform(&t);
		if (p.partitions) s.skip(2);
		transforms[e] = t; // roots keep their world transform; children are overwritten by their local one below
		if (world) world[e] = t;
		if (valid) valid[e] = 1;
	}
	s.pos = p.hierarchy_at;
	for (uint32_t i = 0; i < p.info.n_hierarchy; ++i) {
		const int32_t entity = s.read<int32_t>(), par = s.read<int32_t>();
		s.skip(8); // first_child, next_sibling: implied by the parents
		LmxTransform local;
		read_transform(&local);
		if (entity < 0 || (uint32_t)entity >= n_slots || par >= (int32_t)n_slots) return LMX_ERR_INVALID_ARGUMENT;
		if (par >= 0) { // a Hierarchy record with a parent: the entity is a child, lmx_world_build wants its local transform
			parent[entity] = par;
			transforms[entity] = local;
		}
	}
	return s.overflow ? LMX_ERR_INVALID_ARGUMENT : LMX_OK;
}

} // extern "C"
