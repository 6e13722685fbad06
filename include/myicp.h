/*
 * myicp.h -- drop-in for the reference's ICP/myicp.h:7-36.
 *
 * Same class name, same five public methods with the same signatures and observable behaviour
 * (LoadCloud returns 0; RegisterSymm prints "iters#k / diff: v" per iteration and the final
 * "Result transform / rotation / translation" block, ICP/myicp.cpp:125-126,146-149; defaults
 * max_iters = 10, diff_threshold = 1.0, ICP/myicp.cpp:6), so the reference's main.cpp call
 * sequence (ICP/main.cpp:7-10,16-31) compiles against it unchanged.  The work is done by
 * libsymmicp (include/symmicp.h) on one MI355X: normals (ICP/myicp.cpp:152-172) and the
 * iteration loop (ICP/myicp.cpp:117-142) both run as HIP kernels.
 *
 * Additive surface (BASELINE.json north_star: "align()/setInput*"): setInputSource,
 * setInputTarget, align, getFinalTransformation, and setters for what the reference hard-codes
 * ("todo add params to specify iters & diff", ICP/myicp.h:19).
 */
#pragma once
#include "stdafx.h"
#include "symmicp.h"

typedef pcl::PointXYZ PointT;

class MyICP
{
public:
	MyICP();
	~MyICP();

public:
	int LoadCloud(std::string src_path, std::string tgt_path);
	pcl::PointCloud<PointT>::Ptr GetSrcCloud();
	pcl::PointCloud<PointT>::Ptr GetTgtCloud();

	void RegisterP2P();
	void RegisterSymm();

	// ---- additive ----
	void setInputSource(const float *xyz, const float *normals, size_t n);   // packed [n][3]; normals may be null
	void setInputTarget(const float *xyz, const float *normals, size_t n);
	int align(float out4x4[16] = nullptr, const float *guess4x4 = nullptr);   // returns symmicp_status
	const float *getFinalTransformation() const { return transform_; }       // row-major 4x4
	// the source cloud moved by the final transform (the reference never writes its result back: myicp.cpp:109-111,146-149)
	pcl::PointCloud<PointT>::Ptr GetAlignedSrcCloud() const;
	MyICP(const MyICP &) = delete;
	MyICP &operator=(const MyICP &) = delete;
	void setMaximumIterations(int n) { max_iters = n; }
	void setDiffThreshold(float d) { diff_threshold = d; }
	void setMode(symmicp_mode m) { mode_ = m; }                    // default SYMMICP_MODE_QUIRKS (= the reference)
	void setCorrespondence(symmicp_corr c) { corr_ = c; }          // default SYMMICP_CORR_IDENTITY (= the reference)
	void setVerbose(bool v) { verbose_ = v; }
	const symmicp_result &lastResult() const { return result_; }
	const char *lastError() const { return error_.c_str(); }

private:
	int max_iters;
	float diff_threshold;

	pcl::PointCloud<PointT>::Ptr cloud_src, cloud_tgt;
	pcl::PointCloud<pcl::PointNormal>::Ptr cloud_pn_src, cloud_pn_tgt;

	void estimateNormals();

	symmicp_ctx *context();              // one libsymmicp context for the life of the object (stream, arenas, code objects)

	symmicp_mode mode_;
	symmicp_corr corr_;
	bool verbose_, have_src_normals_, have_tgt_normals_;   // have_*: normals supplied by the caller through setInput*
	symmicp_ctx *ctx_;
	int ctx_corr_;
	float transform_[16];
	symmicp_result result_;
	std::string error_;
};
