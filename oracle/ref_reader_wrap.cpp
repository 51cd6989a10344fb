// ORACLE / TEST INFRASTRUCTURE ONLY (see oracle/lvo.h).  The reference's own dataset readers - /root/reference/include/utils/
// DataReader.hpp (loadImuFile, loadImageList, findFirstAlign: what app/larvioMain.cpp:35-47 calls), compiled where it lies (oracle/Makefile,
// target `ref` -> oracle/_ref/lvref_reader, an executable) against the Eigen stand-in ImuData needs - as a command-line tool that prints
// what it read, for tests/test_oracle_ref_reader.py to set beside the product's examples/lvk_dataset.hpp (through examples/host_tools).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "lvref_eigen2.hpp"
using namespace std;
#include "utils/DataReader.hpp"

int main(int argc, char** argv)
{
    if (argc >= 3 && !strcmp(argv[1], "imu")) {
        vector<larvio::ImuData> v; larvio::loadImuFile(argv[2], v);
        for (auto& d : v) printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", d.timeStampToSec, d.angular_velocity[0], d.angular_velocity[1], d.angular_velocity[2],
                                 d.linear_acceleration[0], d.linear_acceleration[1], d.linear_acceleration[2]);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "images")) {
        vector<larvio::ImgInfo> v; larvio::loadImageList(argv[2], v);
        for (auto& d : v) printf("%.17g [%s]\n", d.timeStampToSec, d.imgName.c_str());
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "align")) {
        vector<larvio::ImuData> a; vector<larvio::ImgInfo> b; larvio::loadImuFile(argv[2], a); larvio::loadImageList(argv[3], b);
        pair<int, int> p(-1, -1); const bool ok = larvio::findFirstAlign(a, b, p);
        printf("%d %d %d\n", ok ? 1 : 0, p.first, p.second);
        return 0;
    }
    return 2;
}
