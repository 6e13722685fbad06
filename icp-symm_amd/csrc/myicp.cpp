// myicp.cpp -- MyICP on top of the libsymmicp C-ABI (drop-in for reference ICP/myicp.cpp:6-172).
#include <cstdio>
#include <cstring>
#include <vector>
#include "myicp.h"

MyICP::MyICP() : max_iters(10), diff_threshold(1.f),                       // myicp.cpp:6
                 mode_(SYMMICP_MODE_QUIRKS), corr_(SYMMICP_CORR_IDENTITY), verbose_(true),
                 have_src_normals_(false), have_tgt_normals_(false), ctx_(nullptr), ctx_corr_(-1)
{
	cloud_src = pcl::PointCloud<PointT>::Ptr(new pcl::PointCloud<PointT>);
	cloud_tgt = pcl::PointCloud<PointT>::Ptr(new pcl::PointCloud<PointT>);
	cloud_pn_src = pcl::PointCloud<pcl::PointNormal>::Ptr(new pcl::PointCloud<pcl::PointNormal>);
	cloud_pn_tgt = pcl::PointCloud<pcl::PointNormal>::Ptr(new pcl::PointCloud<pcl::PointNormal>);
	for (int k = 0; k < 16; k++) transform_[k] = (k % 5 == 0) ? 1.f : 0.f;
	std::memset(&result_, 0, sizeof(result_));
}

MyICP::~MyICP()
{
	if (ctx_) symmicp_destroy(ctx_);
}

// The context is created on first use and kept: a second RegisterSymm / align on the same object reuses its stream, its
// arenas and the loaded code objects.  The correspondence kind decides device layouts, so changing it means a new context.
symmicp_ctx *MyICP::context()
{
	if (ctx_ && ctx_corr_ != (int)corr_) { symmicp_destroy(ctx_); ctx_ = nullptr; }
	if (!ctx_) {
		symmicp_config cfg;
		symmicp_config_default(&cfg);
		cfg.corr = corr_;
		if (symmicp_create(&cfg, &ctx_) != SYMMICP_OK) { ctx_ = nullptr; return nullptr; }
		ctx_corr_ = (int)corr_;
	}
	return ctx_;
}

static bool load_one(const std::string &path, pcl::PointCloud<PointT> &out)
{
	long n = symmicp_pcd_read(path.c_str(), nullptr, nullptr, 0, nullptr);
	out.points.clear();
	if (n < 0) return false;
	std::vector<float> xyz(3 * (size_t)n);
	if (symmicp_pcd_read(path.c_str(), xyz.data(), nullptr, (size_t)n, nullptr) != n) return false;
	out.points.resize((size_t)n);
	for (long i = 0; i < n; i++) { out.points[i].x = xyz[3 * i]; out.points[i].y = xyz[3 * i + 1]; out.points[i].z = xyz[3 * i + 2]; }
	out.width = (uint32_t)n; out.height = 1;
	return true;
}

int MyICP::LoadCloud(std::string src_path, std::string tgt_path)
{
	// myicp.cpp:20-31: x,y,z are kept, other fields dropped; the reference ignores reader status and returns 0
	if (!load_one(src_path, *cloud_src)) error_ = "cannot read " + src_path;
	if (!load_one(tgt_path, *cloud_tgt)) error_ = "cannot read " + tgt_path;
	have_src_normals_ = have_tgt_normals_ = false;
	return 0;
}

pcl::PointCloud<PointT>::Ptr MyICP::GetSrcCloud()
{
	return this->cloud_src;
}

pcl::PointCloud<PointT>::Ptr MyICP::GetTgtCloud()
{
	return this->cloud_tgt;
}

void MyICP::RegisterP2P()
{
	// myicp.cpp:43-59: the reference stub prints an identity guess and applies it (a no-op)
	std::cout << "guess matrix:\n1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1" << std::endl;
}

static void fill_pn(const pcl::PointCloud<PointT> &c, const float *nrm, pcl::PointCloud<pcl::PointNormal> &pn)
{
	pn.points.resize(c.points.size());
	for (size_t i = 0; i < c.points.size(); i++) {
		pcl::PointNormal &p = pn.points[i];
		p.x = c.points[i].x; p.y = c.points[i].y; p.z = c.points[i].z;
		if (nrm) { p.normal_x = nrm[3 * i]; p.normal_y = nrm[3 * i + 1]; p.normal_z = nrm[3 * i + 2]; }
	}
	pn.width = (uint32_t)pn.points.size(); pn.height = 1;
}

void MyICP::setInputSource(const float *xyz, const float *normals, size_t n)
{
	cloud_src->points.resize(n);
	for (size_t i = 0; i < n; i++) { cloud_src->points[i].x = xyz[3 * i]; cloud_src->points[i].y = xyz[3 * i + 1]; cloud_src->points[i].z = xyz[3 * i + 2]; }
	cloud_src->width = (uint32_t)n;
	have_src_normals_ = normals != nullptr;
	fill_pn(*cloud_src, normals, *cloud_pn_src);
}

void MyICP::setInputTarget(const float *xyz, const float *normals, size_t n)
{
	cloud_tgt->points.resize(n);
	for (size_t i = 0; i < n; i++) { cloud_tgt->points[i].x = xyz[3 * i]; cloud_tgt->points[i].y = xyz[3 * i + 1]; cloud_tgt->points[i].z = xyz[3 * i + 2]; }
	cloud_tgt->width = (uint32_t)n;
	have_tgt_normals_ = normals != nullptr;
	fill_pn(*cloud_tgt, normals, *cloud_pn_tgt);
}

void MyICP::estimateNormals()
{
	// myicp.cpp:152-172: NormalEstimation, KdTree, setKSearch(10), viewpoint (0,0,0), then concatenateFields -- from the
	// clouds as they are NOW, on every call (GetSrcCloud / GetTgtCloud hand out the clouds themselves: a caller may have
	// edited them).  Normals the caller supplied through setInput* are kept.
	struct Job { pcl::PointCloud<PointT> *c; pcl::PointCloud<pcl::PointNormal> *pn; bool have; };
	Job jobs[2] = {{cloud_src.get(), cloud_pn_src.get(), have_src_normals_}, {cloud_tgt.get(), cloud_pn_tgt.get(), have_tgt_normals_}};
	for (Job &j : jobs) {
		const size_t n = j.c->points.size();
		if (j.have && j.pn->points.size() == n) {
			for (size_t i = 0; i < n; i++) { j.pn->points[i].x = j.c->points[i].x; j.pn->points[i].y = j.c->points[i].y; j.pn->points[i].z = j.c->points[i].z; }
			continue;
		}
		std::vector<float> nrm(3 * n, 0.f);
		if (n >= 10) {
			symmicp_ctx *ctx = context();
			int st = ctx ? symmicp_ctx_estimate_normals(ctx, &j.c->points[0].x, sizeof(PointT) / sizeof(float), 1, n, 10, nullptr, nrm.data(), nullptr) : SYMMICP_ERR_HIP;
			if (st != SYMMICP_OK) error_ = "symmicp_estimate_normals failed";
		}
		fill_pn(*j.c, nrm.data(), *j.pn);
	}
}

int MyICP::align(float out4x4[16], const float *guess4x4)
{
	assert(cloud_src && cloud_tgt);                                        // myicp.cpp:102
	symmicp_ctx *ctx = context();
	if (!ctx) { error_ = "symmicp_create failed: no usable gfx950 HIP device (there is no CPU fallback)"; result_.status = SYMMICP_ERR_HIP; return SYMMICP_ERR_HIP; }
	estimateNormals();                                                     // myicp.cpp:105
	symmicp_config cfg;
	symmicp_config_default(&cfg);
	cfg.mode = mode_; cfg.corr = corr_; cfg.max_iters = max_iters; cfg.diff_threshold = diff_threshold;
	cfg.verbose = verbose_ ? 1 : 0;
	int st = symmicp_set_config(ctx, &cfg);
	const size_t fs = sizeof(pcl::PointNormal) / sizeof(float);            // pasteInMatrix, func.cpp:5-15
	if (st == SYMMICP_OK) {
		if (!cloud_pn_tgt->points.empty() && !cloud_pn_src->points.empty()) {
			st = symmicp_set_target(ctx, &cloud_pn_tgt->points[0].x, fs, 1, &cloud_pn_tgt->points[0].normal_x, fs, 1, cloud_pn_tgt->points.size());
			if (st == SYMMICP_OK)
				st = symmicp_set_source(ctx, &cloud_pn_src->points[0].x, fs, 1, &cloud_pn_src->points[0].normal_x, fs, 1, cloud_pn_src->points.size());
		} else {
			st = SYMMICP_ERR_SIZE;
		}
	}
	if (st == SYMMICP_OK) st = symmicp_align(ctx, guess4x4, &result_);
	else result_.status = st;
	if (st != SYMMICP_OK) error_ = symmicp_last_error(ctx);
	if (result_.iters > 0 || st == SYMMICP_OK) std::memcpy(transform_, result_.transform, sizeof(transform_));
	if (out4x4) std::memcpy(out4x4, transform_, sizeof(transform_));
	return st;
}

pcl::PointCloud<PointT>::Ptr MyICP::GetAlignedSrcCloud() const
{
	pcl::PointCloud<PointT>::Ptr out(new pcl::PointCloud<PointT>);
	const float *X = transform_;
	out->points.resize(cloud_src->points.size());
	for (size_t i = 0; i < cloud_src->points.size(); i++) {
		const PointT &p = cloud_src->points[i];
		PointT &q = out->points[i];
		q.x = ((X[0] * p.x + X[1] * p.y) + X[2] * p.z) + X[3];
		q.y = ((X[4] * p.x + X[5] * p.y) + X[6] * p.z) + X[7];
		q.z = ((X[8] * p.x + X[9] * p.y) + X[10] * p.z) + X[11];
	}
	out->width = (uint32_t)out->points.size(); out->height = 1;
	return out;
}

void MyICP::RegisterSymm()
{
	// myicp.cpp:100-150; the loop itself (and its stdout lines) runs inside symmicp_align
	int st = align(nullptr, nullptr);
	if (st != SYMMICP_OK) std::cerr << "RegisterSymm: " << error_ << " (status " << st << ")" << std::endl;
}
