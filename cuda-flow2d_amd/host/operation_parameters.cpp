#include "operation_parameters.h"

bool OperationParameters::PushValuePtr(std::string key, void* value_ptr)
{
    return map_.emplace(std::move(key), value_ptr).second;  // existing keys are kept
}

void* OperationParameters::GetValuePtr(std::string key) const
{
    auto it = map_.find(key);
    return it == map_.end() ? nullptr : it->second;
}

void OperationParameters::Clear() { map_.clear(); }
