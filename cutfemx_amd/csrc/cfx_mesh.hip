// cutfemx_amd: background mesh handle (cutcells::MeshView of
// cpp/cutfemx/cut/cut.cpp:500-538) and the synthetic box generator.
#include "cfx_device.h"

using namespace cfx;

namespace
{

__global__ void compact_conn_kernel(const int32_t* __restrict__ in, int64_t ncells, int width, int stride,
                                    int32_t* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncells * width) return;
  const int64_t c = i / width;
  const int k = (int)(i - c * width);
  out[i] = in[c * stride + k];
}

// box [0,1]^d, n^d cubes; vertex id ix+(n+1)(iy+(n+1)iz)
__global__ void box_nodes_kernel(int tdim, int n, int z0, int64_t nnodes, double* __restrict__ x)
{
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nnodes) return;
  const int64_t n1 = n + 1;
  const int64_t ix = v % n1, iy = (v / n1) % n1, iz = tdim == 3 ? v / (n1 * n1) : 0;
  x[3 * v + 0] = (double)ix / (double)n;
  x[3 * v + 1] = (double)iy / (double)n;
  x[3 * v + 2] = tdim == 3 ? (double)(iz + z0) / (double)n : 0.0;
}

// Kuhn split; local corner i = bx + 2 by + 4 bz
// (cpp/cutfemx/distance/fast_iterative.h:93-94,103-108)
__constant__ int c_kuhn_tet[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};
__constant__ int c_kuhn_tri[2][3] = {{0, 1, 3}, {0, 3, 2}};

__global__ void box_cells_kernel(int tdim, int n, int64_t ncells, int32_t* __restrict__ conn)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncells) return;
  const int64_t n1 = n + 1;
  if (tdim == 2)
  {
    const int64_t q = c / 2;
    const int k = (int)(c - 2 * q);
    const int64_t ix = q % n, iy = q / n;
    for (int j = 0; j < 3; ++j)
    {
      const int i = c_kuhn_tri[k][j];
      conn[c * 3 + j] = (int32_t)((ix + (i & 1)) + n1 * (iy + ((i >> 1) & 1)));
    }
    return;
  }
  const int64_t h = c / 6;
  const int k = (int)(c - 6 * h);
  const int64_t ix = h % n, iy = (h / n) % n, iz = h / ((int64_t)n * n);
  int32_t v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    const int i = c_kuhn_tet[k][j];
    v[j] = (int32_t)((ix + (i & 1)) + n1 * ((iy + ((i >> 1) & 1)) + n1 * (iz + ((i >> 2) & 1))));
  }
  *reinterpret_cast<int4*>(conn + c * 4) = make_int4(v[0], v[1], v[2], v[3]);
}

} // namespace

extern "C" {

int cfx_mesh_create(int tdim, int gdim, int64_t nnodes, const double* x, int64_t ncells,
                    const int32_t* conn, int cell_stride, cfx_mesh_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create: null output");
  require(tdim == 2 || tdim == 3, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create: tdim must be 2 or 3 (simplex cells)");
  require(gdim == tdim, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create: gdim must equal tdim");
  require(nnodes > 0 && ncells >= 0 && x && (conn || ncells == 0), CFX_ERR_INVALID_ARGUMENT,
          "cfx_mesh_create: empty mesh arrays");
  require(ncells < 2147483647LL && nnodes < 2147483647LL, CFX_ERR_INVALID_ARGUMENT,
          "cfx_mesh_create: entity counts must fit int32");
  const int nv = tdim + 1;
  require(cell_stride >= nv, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create: cell_stride < vertices per cell");
  auto m = std::make_unique<cfx_mesh_s>();
  m->tdim = tdim; m->gdim = gdim; m->nnodes = nnodes; m->ncells = ncells;
  m->x = to_device(x, nnodes * 3);
  if (cell_stride == nv)
    m->conn = to_device_aligned(conn, ncells * nv);
  else
  {
    DevArray<int32_t> raw = to_device(conn, ncells * (int64_t)cell_stride);
    m->conn.alloc(ncells * nv);
    launch("compact_conn", compact_conn_kernel, grid_for(ncells * nv), dim3(kBlock), 0, raw.p, ncells, nv,
           cell_stride, m->conn.p);
    CFX_HIP(hipStreamSynchronize(ctx().stream));
  }
  *out = m.release();
  CFX_API_END
}

int cfx_mesh_create_box(int tdim, int n, cfx_mesh_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_box: null output");
  require(tdim == 2 || tdim == 3, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_box: tdim must be 2 or 3");
  require(n >= 1, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_box: n must be >= 1");
  int64_t nnodes = 1, ncubes = 1;
  for (int d = 0; d < tdim; ++d) { nnodes *= (n + 1); ncubes *= n; }
  const int64_t ncells = ncubes * (tdim == 2 ? 2 : 6);
  require(ncells < 2147483647LL, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_box: too many cells for int32 ids");
  auto m = std::make_unique<cfx_mesh_s>();
  m->tdim = tdim; m->gdim = tdim; m->nnodes = nnodes; m->ncells = ncells;
  m->x.alloc(nnodes * 3);
  m->conn.alloc(ncells * (tdim + 1));
  m->box_n = n;
  launch("box_nodes", box_nodes_kernel, grid_for(nnodes), dim3(kBlock), 0, tdim, n, 0, nnodes, m->x.p);
  launch("box_cells", box_cells_kernel, grid_for(ncells), dim3(kBlock), 0, tdim, n, ncells, m->conn.p);
  *out = m.release();
  CFX_API_END
}

int cfx_mesh_create_slab(int n, int z0, int nz, cfx_mesh_t* out)
{
  CFX_API_BEGIN
  ctx().ensure();
  require(out != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_slab: null output");
  require(n >= 1 && z0 >= 0 && nz >= 1 && z0 + nz <= n, CFX_ERR_INVALID_ARGUMENT,
          "cfx_mesh_create_slab: need 0 <= z0, 1 <= nz, z0 + nz <= n");
  const int64_t n1 = n + 1;
  const int64_t nnodes = n1 * n1 * (nz + 1), ncells = 6LL * n * n * nz;
  require(ncells < 2147483647LL, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_create_slab: too many cells for int32 ids");
  auto m = std::make_unique<cfx_mesh_s>();
  m->tdim = 3; m->gdim = 3; m->nnodes = nnodes; m->ncells = ncells;
  m->x.alloc(nnodes * 3);
  m->conn.alloc(ncells * 4);
  // the same generators as the full box: local ids = global ids minus the slab offset
  m->box_n = n;
  launch("box_nodes", box_nodes_kernel, grid_for(nnodes), dim3(kBlock), 0, 3, n, z0, nnodes, m->x.p);
  launch("box_cells", box_cells_kernel, grid_for(ncells), dim3(kBlock), 0, 3, n, ncells, m->conn.p);
  *out = m.release();
  CFX_API_END
}

int cfx_mesh_info(cfx_mesh_t m, int* tdim, int* gdim, int64_t* nnodes, int64_t* ncells, const double** x,
                  const int32_t** conn)
{
  CFX_API_BEGIN
  require(m != nullptr, CFX_ERR_INVALID_ARGUMENT, "cfx_mesh_info: null mesh");
  if (tdim) *tdim = m->tdim;
  if (gdim) *gdim = m->gdim;
  if (nnodes) *nnodes = m->nnodes;
  if (ncells) *ncells = m->ncells;
  if (x) *x = m->x.p;
  if (conn) *conn = m->conn.p;
  CFX_API_END
}

int cfx_mesh_destroy(cfx_mesh_t m)
{
  CFX_API_BEGIN
  delete m;
  CFX_API_END
}

} // extern "C"
