// lmx_world_blob.cpp — host-side ingest of a serialized World (engine/world.cpp:837-1043) into the inputs of lmx_world_build:
// header, module list, flags, the LZ4 block (Engine::compress = LZ4 block format, engine.cpp:254-269), entity transforms,
// entity names (skipped), hierarchy records. Module payloads that follow are not read. Pure host code, no device needed.
#include <cstdint>
#include <cstring>
#include <new>
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
	for (int32_t i = 0; i < n_modules; ++i)
		if (!in.skip_string()) return LMX_ERR_INVALID_ARGUMENT;
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

} // namespace

extern "C" {

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
