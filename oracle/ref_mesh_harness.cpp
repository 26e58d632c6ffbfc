/*
 * ref_mesh_harness.cpp -- thin C entry points around the REFERENCE's own host mesh code.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libkfrefmesh.so, in this container only, from the
 * sources where they lie under /root/reference/src (nothing is copied into the repo):
 *   utils/mesh/meshData.h + meshData.cpp   ml::MeshData<float>: mergeCloseVertices / removeDegeneratedFaces /
 *                                          removeDuplicateFaces / computeVertexNormals
 *   utils/mesh/MeshIO.h + MeshIO.cpp       ml::MeshIO<float>::saveToFile (OBJ / PLY / OFF writers)
 *   utils/mesh/sparseGrid3.h, cuda/cuda_declar.h (float3 operators)
 * They depend only on the C++ standard library and the CUDA vector-type *headers* that ship in this image (triton wheel).
 *
 * ref_mesh_build() replays MeshGeneratorMarchingcube::copyTrianglesToCPU + saveMesh
 * (/root/reference/src/MeshGeneratorMarchingcube.cpp:30-96) statement by statement on a triangle soup handed in by the
 * caller (the marching-cubes output, layout of kf_triangle: 3 x {pos[3], color[3]}), calling the reference's own methods.
 * tools/make_mesh_golden.py turns its outputs into tests/golden/mesh_*.npz, which pin hybkinectfu_amd/host's MeshData
 * (index for index, bit for bit) and its file writers (byte for byte).
 */
#include <cstring>
#include <string>
#include "cuda/cuda_declar.h"
#include "utils/mesh/MeshIO.h"

namespace {
struct Vtx { float pos[3]; float color[3]; };
struct Tri { Vtx v[3]; };
}

extern "C" {

void* ref_mesh_build(const void* triangles, unsigned triangle_num, int use_rgb) {
  const Tri* tris = (const Tri*)triangles;
  ml::MeshDataf* m = new ml::MeshDataf();
  ml::MeshDataf& _meshes = *m;
  // copyTrianglesToCPU :39-58
  _meshes.m_Vertices.resize(3 * triangle_num);
  if (use_rgb) _meshes.m_Colors.resize(3 * triangle_num);
  for (unsigned i = 0; i < triangle_num; ++i) {
    for (int k = 0; k < 3; ++k) {
      const Vtx& v = tris[i].v[k];
      _meshes.m_Vertices[3 * i + k] = make_float3(v.pos[0], v.pos[1], v.pos[2]);
      if (use_rgb) _meshes.m_Colors[3 * i + k] = make_float4(v.color[2], v.color[1], v.color[0], 1.0);      // :53 z,y,x
    }
  }
  // saveMesh :69-84 -- including the index buffer sized by the VERTEX count (:70): the surplus faces are (0,0,0)
  _meshes.m_FaceIndicesVertices.resize(_meshes.m_Vertices.size());
  for (unsigned int i = 0; i < (unsigned int)_meshes.m_Vertices.size() / 3; i++) {
    _meshes.m_FaceIndicesVertices[i][0] = 3 * i + 0;
    _meshes.m_FaceIndicesVertices[i][1] = 3 * i + 1;
    _meshes.m_FaceIndicesVertices[i][2] = 3 * i + 2;
  }
  _meshes.mergeCloseVertices(0.0001f, true);
  _meshes.removeDuplicateFaces();
  _meshes.computeVertexNormals();
  return m;
}
void ref_mesh_destroy(void* h) { delete (ml::MeshDataf*)h; }
unsigned ref_mesh_vertex_count(void* h) { return (unsigned)((ml::MeshDataf*)h)->m_Vertices.size(); }
unsigned ref_mesh_face_count(void* h) { return (unsigned)((ml::MeshDataf*)h)->m_FaceIndicesVertices.size(); }
unsigned ref_mesh_color_count(void* h) { return (unsigned)((ml::MeshDataf*)h)->m_Colors.size(); }
/* out_faces: 3 indices per face; returns the number of faces whose valence is not 3 (must be 0 for a triangle soup) */
unsigned ref_mesh_read(void* h, float* out_vertices, float* out_normals, float* out_colors, unsigned* out_faces) {
  ml::MeshDataf& m = *(ml::MeshDataf*)h;
  for (size_t i = 0; i < m.m_Vertices.size(); ++i) { out_vertices[3 * i] = m.m_Vertices[i].x; out_vertices[3 * i + 1] = m.m_Vertices[i].y; out_vertices[3 * i + 2] = m.m_Vertices[i].z; }
  for (size_t i = 0; i < m.m_Normals.size(); ++i) { out_normals[3 * i] = m.m_Normals[i].x; out_normals[3 * i + 1] = m.m_Normals[i].y; out_normals[3 * i + 2] = m.m_Normals[i].z; }
  if (out_colors) for (size_t i = 0; i < m.m_Colors.size(); ++i) { out_colors[4 * i] = m.m_Colors[i].x; out_colors[4 * i + 1] = m.m_Colors[i].y; out_colors[4 * i + 2] = m.m_Colors[i].z; out_colors[4 * i + 3] = m.m_Colors[i].w; }
  unsigned odd = 0;
  for (unsigned i = 0; i < m.m_FaceIndicesVertices.size(); ++i) {
    const auto& f = m.m_FaceIndicesVertices[i];
    if (f.size() != 3) { ++odd; out_faces[3 * i] = out_faces[3 * i + 1] = out_faces[3 * i + 2] = 0xffffffffu; continue; }
    out_faces[3 * i] = f[0]; out_faces[3 * i + 1] = f[1]; out_faces[3 * i + 2] = f[2];
  }
  return odd;
}
/* saveMesh :92 */
void ref_mesh_save(void* h, const char* filename) { ml::MeshIOf::saveToFile(std::string(filename), *(ml::MeshDataf*)h); }

}
