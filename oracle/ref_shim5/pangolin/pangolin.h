// ORACLE / TEST INFRASTRUCTURE ONLY.  A headless pangolin + OpenGL for the reference's driver (app/larvioMain.cpp:57-78,119-199 and
// include/visualization/visualize.hpp): every drawing call is a no-op, ShouldQuit() is true (the driver's closing loop, larvioMain.cpp:179-199, is skipped), and the
// one thing a test can look at - the body pose the driver hands to OpenGlRenderState::Follow after every odometry update
// (larvioMain.cpp:122-133: GetCurrentOpenGLPoseMatrix(Tbw_pgl, Estimator->getTbw()), column-major 4 x 4) - is appended to the file
// LVREF_MAIN_POSES names: one line per call, 16 numbers with 17 significant digits.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <sstream>       // (the real pangolin.h brings it; larvioMain.cpp:156 relies on that)
#include <cstring>
#include <string>
#define GL_DEPTH_TEST 0
#define GL_BLEND 0
#define GL_SRC_ALPHA 0
#define GL_ONE_MINUS_SRC_ALPHA 0
#define GL_COLOR_BUFFER_BIT 0
#define GL_DEPTH_BUFFER_BIT 0
#define GL_LINES 0
#define GL_POINTS 0
#define GL_BGR 0
#define GL_UNSIGNED_BYTE 0
inline void glEnable(int) {}
inline void glBlendFunc(int, int) {}
inline void glClear(int) {}
inline void glClearColor(float, float, float, float) {}
inline void glPushMatrix() {}
inline void glPopMatrix() {}
inline void glMultMatrixd(const double*) {}
inline void glLineWidth(float) {}
inline void glPointSize(float) {}
inline void glColor3f(float, float, float) {}
inline void glBegin(int) {}
inline void glEnd() {}
inline void glVertex3f(float, float, float) {}
namespace pangolin {
struct OpenGlMatrix { double m[16]; OpenGlMatrix() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0; } };
struct Attach { double v; Attach(double x = 0) : v(x) {} static Attach Pix(int p) { return Attach((double)p); } };
enum Lock { LockLeft, LockBottom };
inline void CreateWindowAndBind(const std::string&, int, int) {}
struct Handler3D;
struct View {
    View& SetBounds(Attach, Attach, Attach, Attach) { return *this; }
    View& SetBounds(Attach, Attach, Attach, Attach, double) { return *this; }
    View& SetHandler(Handler3D*) { return *this; }
    View& SetLock(Lock, Lock) { return *this; }
    template <typename S> void Activate(const S&) {}
    void Activate() {}
};
inline View& CreatePanel(const std::string&) { static View v; return v; }
inline View& CreateDisplay() { static View v[4]; static int k = 0; return v[(k++) & 3]; }
template <typename T> struct Var { T v; Var(const std::string&, T init, bool = false) : v(init) {} operator T() const { return v; } };
inline OpenGlMatrix ProjectionMatrix(int, int, double, double, double, double, double, double) { return OpenGlMatrix(); }
inline OpenGlMatrix ModelViewLookAt(double, double, double, double, double, double, double, double, double) { return OpenGlMatrix(); }
struct OpenGlRenderState {
    OpenGlRenderState(const OpenGlMatrix&, const OpenGlMatrix&) {}
    void SetModelViewMatrix(const OpenGlMatrix&) {}
    void Follow(const OpenGlMatrix& T)
    {
        const char* p = std::getenv("LVREF_MAIN_POSES"); if (!p) return;
        FILE* f = std::fopen(p, "a"); if (!f) return;
        for (int i = 0; i < 16; ++i) std::fprintf(f, "%.17g%c", T.m[i], i == 15 ? '\n' : ' ');
        std::fclose(f);
    }
};
struct Handler3D { Handler3D(OpenGlRenderState&) {} };
struct GlTexture { GlTexture(int, int) {} void Upload(const void*, int, int) {} void RenderToViewportFlipY() {} };
inline void FinishFrame() {}
inline bool ShouldQuit() { return true; }
}
