// examples/main.cpp -- the reference driver (ICP/main.cpp:5-41) against the MI355X engine.
// Same call sequence: construct, LoadCloud("cat.pcd","cat_out.pcd"), RegisterSymm(), read the clouds back
// point by point, hand the target to a viewer (a no-op here), spin until the viewer stops.
// File names may be given on the command line; the defaults are the reference's (ICP/main.cpp:8).
#include "stdafx.h"

#include "myicp.h"

int main(int argc, char** argv)
{
	MyICP myicp;
	myicp.LoadCloud(argc > 1 ? argv[1] : "cat.pcd", argc > 2 ? argv[2] : "cat_out.pcd");

	myicp.RegisterSymm();

	// visualize
	pcl::visualization::PCLVisualizer viewer("test");
	// translate to PointXYZ type so that can be added into viewer
	pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1(new pcl::PointCloud<pcl::PointXYZ>);
	pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2(new pcl::PointCloud<pcl::PointXYZ>);
	pcl::PointCloud<PointT>::Ptr cloud = myicp.GetSrcCloud();
	int npts = cloud->points.size();
	for (size_t i = 0; i < npts; i++)
	{
		pcl::PointXYZ p;
		p.x = cloud->points[i].x, p.y = cloud->points[i].y, p.z = cloud->points[i].z;
		cloud1->points.push_back(p);
	}
	cloud = myicp.GetTgtCloud();
	npts = cloud->points.size();
	for (size_t i = 0; i < npts; i++)
	{
		pcl::PointXYZ p;
		p.x = cloud->points[i].x, p.y = cloud->points[i].y, p.z = cloud->points[i].z;
		cloud2->points.push_back(p);
	}
	viewer.addPointCloud(cloud2);

	// show viewer
	while (!viewer.wasStopped())
	{
		viewer.spinOnce();
	}
	return myicp.lastResult().status;
}
