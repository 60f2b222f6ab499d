/* TEST INFRASTRUCTURE. Dumps what the reference's own scene reader (ext/libvkr/src/vkr.c, compiled unmodified from the
 * reference checkout into oracle/_ref/libvkr_ref.so together with this file) makes of a .vks file, as JSON, so that
 * tests can hold realtimepathtracingresearchframework_amd/vks.py against it. Only calls libvkr's public API (vkr.h);
 * floats are printed as their bit patterns so that comparisons are exact. */
#include "vkr.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

static unsigned bits(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
static void put_tex(FILE *o, const char *key, const VkrTexture *t) {
    if (!t->filename) {
        fprintf(o, "\"%s\": null", key);
        return;
    }
    fprintf(o, "\"%s\": {\"width\": %d, \"height\": %d, \"format\": %d, \"numMipLevels\": %d, \"dataSize\": %llu, \"dataOffset\": %lld}", key,
            t->width, t->height, t->format, t->numMipLevels, (unsigned long long)t->dataSize, (long long)t->dataOffset);
}
static char g_err[1024];
static void on_error(VkrResult r, const char *msg) {
    (void)r;
    strncpy(g_err, msg, sizeof(g_err) - 1);
}
const char *ref_vkr_last_error(void) { return g_err; }

int ref_vkr_dump(const char *vks_path, const char *json_path) {
    VkrScene v;
    g_err[0] = 0;
    VkrResult r = vkr_open_scene(vks_path, &v, on_error);
    if (r != VKR_SUCCESS) return (int)r;
    FILE *o = fopen(json_path, "w");
    if (!o) {
        vkr_close_scene(&v);
        return -100;
    }
    fprintf(o, "{\"version\": %d, \"flags\": %u, \"headerSize\": %lld, \"dataOffset\": %lld, \"numMaterials\": %llu, \"numTriangles\": %llu,\n", v.version,
            v.flags, (long long)v.headerSize, (long long)v.dataOffset, (unsigned long long)v.numMaterials, (unsigned long long)v.numTriangles);
    fprintf(o, " \"numMeshes\": %llu, \"numInstances\": %llu, \"numLodGroups\": %llu, \"numFrames\": %llu, \"numStaticTransforms\": %llu,\n",
            (unsigned long long)v.numMeshes, (unsigned long long)v.numInstances, (unsigned long long)v.numLodGroups, (unsigned long long)v.numFrames,
            (unsigned long long)v.numStaticTransforms);
    fprintf(o, " \"numAnimatedTransforms\": %llu, \"animationOffset\": %lld, \"textureDir\": \"%s\",\n", (unsigned long long)v.numAnimatedTransforms,
            (long long)v.animationOffset, v.textureDir ? v.textureDir : "");
    fprintf(o, " \"meshes\": [\n");
    for (uint64_t i = 0; i < v.numMeshes; ++i) {
        const VkrMesh *m = v.meshes + i;
        fprintf(o, "  {\"name\": \"%s\", \"vertexScale\": [%u, %u, %u], \"vertexOffset\": [%u, %u, %u], \"flags\": %u, \"numSegments\": %llu,", m->name,
                bits(m->vertexScale[0]), bits(m->vertexScale[1]), bits(m->vertexScale[2]), bits(m->vertexOffset[0]), bits(m->vertexOffset[1]),
                bits(m->vertexOffset[2]), m->flags, (unsigned long long)m->numSegments);
        fprintf(o, " \"materialIdBufferBase\": %d, \"numMaterialsInRange\": %u, \"numTriangles\": %llu, \"lodGroup\": %lld,", m->materialIdBufferBase,
                m->numMaterialsInRange, (unsigned long long)m->numTriangles, (long long)m->lodGroup);
        fprintf(o, " \"vertexBufferOffset\": %lld, \"normalUvBufferOffset\": %lld, \"materialIdBufferOffset\": %lld, \"materialIdSize\": %d,",
                (long long)m->vertexBufferOffset, (long long)m->normalUvBufferOffset, (long long)m->materialIdBufferOffset, (int)m->materialIdSize);
        fprintf(o, " \"indexBufferOffset\": %lld, \"segmentNumTriangles\": [", (long long)m->indexBufferOffset);
        for (uint64_t j = 0; j < m->numSegments; ++j) fprintf(o, "%s%llu", j ? ", " : "", (unsigned long long)m->segmentNumTriangles[j]);
        fprintf(o, "], \"segmentMaterialBaseOffsets\": [");
        for (uint64_t j = 0; j < m->numSegments; ++j) fprintf(o, "%s%d", j ? ", " : "", m->segmentMaterialBaseOffsets[j]);
        fprintf(o, "]}%s\n", i + 1 < v.numMeshes ? "," : "");
    }
    fprintf(o, " ],\n \"instances\": [\n");
    for (uint64_t i = 0; i < v.numInstances; ++i) {
        const VkrInstance *in = v.instances + i;
        fprintf(o, "  {\"name\": \"%s\", \"meshId\": %lld, \"transformIndex\": %u, \"flags\": %u}%s\n", in->name, (long long)in->meshId, in->transformIndex,
                in->flags, i + 1 < v.numInstances ? "," : "");
    }
    fprintf(o, " ],\n \"lodGroups\": [");
    for (uint64_t i = 0; i < v.numLodGroups; ++i) {
        const VkrLodGroup *g = v.lodGroups + i;
        fprintf(o, "%s{\"numLevelsOfDetail\": %llu, \"meshIds\": [", i ? ", " : "", (unsigned long long)g->numLevelsOfDetail);
        for (uint64_t j = 0; j < g->numLevelsOfDetail; ++j) fprintf(o, "%s%lld", j ? ", " : "", (long long)g->meshIds[j]);
        fprintf(o, "], \"detailReduction\": [");
        for (uint64_t j = 0; j < g->numLevelsOfDetail; ++j) fprintf(o, "%s%u", j ? ", " : "", bits(g->detailReduction[j]));
        fprintf(o, "]}");
    }
    fprintf(o, "],\n \"materials\": [\n");
    for (uint64_t i = 0; i < v.numMaterials; ++i) {
        const VkrMaterial *m = v.materials + i;
        fprintf(o, "  {\"name\": \"%s\", \"emissionIntensity\": %u, \"emitterBaseColor\": [%u, %u, %u], \"specularTransmission\": %u, \"iorEta\": %u,", m->name,
                bits(m->emissionIntensity), bits(m->emitterBaseColor[0]), bits(m->emitterBaseColor[1]), bits(m->emitterBaseColor[2]),
                bits(m->specularTransmission), bits(m->iorEta));
        fprintf(o, " \"iorK\": %u, \"translucency\": %u, ", bits(m->iorK), bits(m->translucency));
        put_tex(o, "texBaseColor", &m->texBaseColor);
        fprintf(o, ", ");
        put_tex(o, "texNormal", &m->texNormal);
        fprintf(o, ", ");
        put_tex(o, "texSpecular", &m->texSpecularRoughnessMetalness);
        fprintf(o, "}%s\n", i + 1 < v.numMaterials ? "," : "");
    }
    /* the static transforms, dequantised by the reference (file versions < 4 keep the table in memory, 4 in the file) */
    fprintf(o, " ],\n \"transforms\": [\n");
    unsigned char *table = v.animationData;
    unsigned char *loaded = NULL;
    if (!table && v.animationOffset > 0) {
        FILE *f = fopen(vks_path, "rb");
        size_t n = (size_t)v.numStaticTransforms * VKR_QUANTIZED_TRANSFORM_SIZE;
        loaded = (unsigned char *)malloc(n ? n : 1);
        if (f && loaded && fseek(f, (long)v.animationOffset, SEEK_SET) == 0 && fread(loaded, 1, n, f) == n) table = loaded;
        if (f) fclose(f);
    }
    for (uint64_t i = 0; table && i < v.numStaticTransforms; ++i) {
        float m[4][3];
        vkr_dequantize_transform(m, table + vkr_get_transform_offset((uint32_t)i, v.numStaticTransforms, v.numAnimatedTransforms, 0) * VKR_QUANTIZED_TRANSFORM_SIZE);
        fprintf(o, "  [");
        for (int k = 0; k < 12; ++k) fprintf(o, "%s%u", k ? ", " : "", bits(m[k / 3][k % 3]));
        fprintf(o, "]%s\n", i + 1 < v.numStaticTransforms ? "," : "");
    }
    fprintf(o, " ]}\n");
    free(loaded);
    fclose(o);
    vkr_close_scene(&v);
    return 0;
}
