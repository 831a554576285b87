// Mirrors the two accessors nodes use: RHI::Renderer::GetDriver() / GetDriverCommands() (RHI/Renderer.cpp:60-63,156-164).
// Backend selection is compile-time in the reference (SAILOR_BUILD_WITH_VULKAN); this mirror is the SAILOR_BUILD_WITH_HIP case.
#pragma once
#include <memory>
#include "GraphicsDriver.h"

namespace Sailor::RHI {

class Renderer {
public:
    // `stream` = a hipStream_t owned by the caller (nullptr = default stream); ownStream = create a dedicated one
    Renderer(int deviceOrdinal, void* stream, bool ownStream);
    ~Renderer();
    static IGraphicsDriver* GetDriver() { return s_driver; }
    static IGraphicsDriverCommands* GetDriverCommands() { return s_commands; }
    int GetStatus() const { return m_status; }
private:
    std::unique_ptr<IGraphicsDriver> m_driverInstance;
    static IGraphicsDriver* s_driver;
    static IGraphicsDriverCommands* s_commands;
    int m_status = 0;
};

} // namespace Sailor::RHI
